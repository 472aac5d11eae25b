"""Drop-in check at the test level: the reference's OWN host-side test functions, executed unmodified from the files
where they lie, with `bitblas_amd` answering to the name `bitblas`.

Only tests that need no GPU run here (operator construction + planning + `get_source`, config hashing, operator
cache): the reference checkout does not exist on the GPU box, so nothing under `-m gpu` may depend on it - the
forward-pass expectations of the same files travel as fixtures instead (tests/golden/optest_golden.*).
Skipped when /root/reference is absent.  Nothing is copied: each file is compiled from its path at run time."""
import os
import sys
import types

import pytest

import bitblas_amd

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "testing", "python")),
                                reason="reference checkout not present")


@pytest.fixture
def as_bitblas():
    """`bitblas_amd.install_as_bitblas()` (bitblas_amd/compat.py) + inert stand-ins for the code-generator names the test
    modules import at their top but the selected tests never use (`tvm`, `bitblas.tl.lower`); everything is removed from
    sys.modules afterwards"""
    from bitblas_amd import compat
    added = {}

    def put(name, mod):
        added[name] = sys.modules.get(name)
        sys.modules[name] = mod

    shim = bitblas_amd.install_as_bitblas()
    tvm = types.ModuleType("tvm")
    tvm.__path__ = []
    contrib = types.ModuleType("tvm.contrib")
    contrib.__path__ = []
    cutils = types.ModuleType("tvm.contrib.utils")
    contrib.utils = cutils
    tvm.contrib = contrib
    shim.tvm = tvm
    tl = types.ModuleType("bitblas.tl")
    tl.__path__ = []
    lower = types.ModuleType("bitblas.tl.lower")
    lower.tl_lower = None
    tl.lower = lower
    shim.tl = tl
    put("bitblas.tl", tl)
    put("bitblas.tl.lower", lower)
    put("tvm", tvm)
    put("tvm.contrib", contrib)
    put("tvm.contrib.utils", cutils)
    yield shim
    for name, old in added.items():
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
    compat.uninstall()


def load(relpath):
    path = os.path.join(REF, "testing", "python", relpath)
    ns = {"__name__": "reference_test_module", "__file__": path}
    exec(compile(open(path).read(), path, "exec"), ns)
    return ns


def run_parametrized(fn):
    """call a @pytest.mark.parametrize'd reference test once per parameter tuple"""
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    assert len(marks) == 1
    names = [n.strip() for n in marks[0].args[0].split(",")]
    n = 0
    for values in marks[0].args[1]:
        fn(**dict(zip(names, values)))
        n += 1
    return n


@pytest.mark.parametrize("relpath", ["operators/test_general_matmul_ops_backend_tl.py",
                                     "operators/test_general_matmul_ops_backend.py"])
def test_reference_codegen_and_finetune_tests_pass_against_this_package(as_bitblas, relpath):
    """`test_matmul_codegen_default` (12 configurations each: fp16, int8 x int8, uint4 +- scale / zeros / bias,
    M = 1 and 768) and `test_matmul_finetune` (static and dynamic M): construct the operator, tune, ask for the
    source - backend_tl.py:291-325, backend.py:180-209."""
    ns = load(relpath)
    if relpath.endswith("backend_tl.py"):
        ns["test_matmul_codegen_default"]()    # the tir-backend twin lowers `matmul.prim_func` through TVM itself
    ns["test_matmul_finetune"]()


def test_reference_operator_cache_tests_pass_against_this_package(as_bitblas):
    """cache/test_operator_cache.py:22-135: config hashing and global_operator_cache add / get, static and dynamic M"""
    ns = load("cache/test_operator_cache.py")
    assert run_parametrized(ns["test_config_hashable"]) == 3
    assert run_parametrized(ns["test_global_cache_inquery"]) == 3


def test_reference_cache_spin_lock_threads_pass_against_this_package(as_bitblas, tmp_path):
    """cache/test_operator_cache_spin_lock.py:21-46: the reference's own worker - construct, add to the global cache, save the
    database, clear, reload - run from four threads at once on one database path, exactly as its test does (:87-98); the
    test's tail (:100-122) is a GPU forward and travels as tests/test_gemv_gpu.py instead."""
    import threading
    ns = load("cache/test_operator_cache_spin_lock.py")
    cfg = bitblas_amd.MatmulConfig(M=1, N=1024, K=1024, A_dtype="float16", out_dtype="float16", accum_dtype="float16",
                                   with_bias=False, propagate_a=False, propagate_b=False, layout="nt")
    bitblas_amd.global_operator_cache.clear()
    errors = []

    def worker(i):
        try:
            ns["tune_op_in_thread"](i, cfg, str(tmp_path))
        except BaseException as exc:  # noqa: BLE001 - an assertion in a thread must fail the test
            errors.append((i, exc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    try:
        assert not errors, errors
        op = bitblas_amd.global_operator_cache.get(cfg)
        assert isinstance(op, bitblas_amd.Matmul) and op.config == cfg
    finally:
        bitblas_amd.global_operator_cache.clear()
