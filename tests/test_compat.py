"""`bitblas_amd.install_as_bitblas()`: the reference's import paths answered by this package (bitblas_amd/compat.py).

CPU only.  The second half executes the reference's own caller module - integration/BitNet/utils_quant.py, unmodified,
from the file where it lies - against the alias; skipped where the checkout is absent (the GPU box)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import bitblas_amd
from bitblas_amd import compat

REF = "/root/reference"


@pytest.fixture
def bitblas_alias():
    assert "bitblas" not in sys.modules, "a real bitblas is imported in this process"
    top = bitblas_amd.install_as_bitblas()
    yield top
    compat.uninstall()
    assert "bitblas" not in sys.modules and "bitblas.ops.general_matmul" not in sys.modules
    assert not any(isinstance(f, compat._RefusingFinder) for f in sys.meta_path)


def test_import_paths_of_the_callers(bitblas_alias):
    """every import statement the reference's callers and tests use for this path (SURVEY.md section 8b "Callers")"""
    ns = {}
    exec("import bitblas\n"
         "import bitblas.testing\n"
         "from bitblas import Matmul, MatmulConfig, Linear, auto_detect_nvidia_target, set_log_level\n"
         "from bitblas import MatmulConfigWithSplitK, MatmulWithSplitK\n"
         "from bitblas.ops import Operator, OperatorConfig, Matmul as M2, MatmulConfig as C2\n"
         "from bitblas.ops.general_matmul import Matmul as M3, MatmulConfig as C3, is_native_compute\n"
         "from bitblas.ops.general_matmul_splitk import MatmulConfigWithSplitK as CS, MatmulWithSplitK as MS\n"
         "from bitblas.ops.operator import OPExecutorCPU, TransformKind\n"
         "from bitblas.ops.common import OptimizeStrategy\n"
         "from bitblas.cache import global_operator_cache, get_database_path, OperatorCache, load_global_ops_cache\n"
         "from bitblas.cache.operator import OperatorCache as OC2\n"
         "from bitblas.module import Linear as L2, unpack_qweight, unpack_qzeros\n"
         "from bitblas.quantization.utils import general_compress, interleave_weight, gen_quant4\n"
         "from bitblas.quantization import general_compress as gc2\n"
         "from bitblas.utils import auto_detect_nvidia_target as det2\n"
         "from bitblas.utils.target_detector import auto_detect_nvidia_target as det3\n"
         "from bitblas.common import BITBLAS_DEFAULT_CACHE_PATH\n", ns)
    assert ns["Matmul"] is ns["M2"] is ns["M3"] is bitblas_amd.Matmul
    assert ns["MatmulConfig"] is ns["C2"] is ns["C3"] is bitblas_amd.MatmulConfig
    assert ns["CS"] is bitblas_amd.MatmulConfigWithSplitK and ns["MS"] is bitblas_amd.MatmulWithSplitK
    assert ns["Linear"] is ns["L2"] is bitblas_amd.Linear
    assert ns["global_operator_cache"] is bitblas_amd.global_operator_cache and ns["OC2"] is bitblas_amd.OperatorCache
    assert ns["det2"] is ns["det3"] is bitblas_amd.auto_detect_nvidia_target
    assert ns["gc2"] is ns["general_compress"] is bitblas_amd.general_compress
    assert issubclass(ns["Matmul"], ns["Operator"]) and issubclass(ns["MS"], ns["Matmul"])
    assert ns["BITBLAS_DEFAULT_CACHE_PATH"].endswith(os.path.join(".cache", "bitblas"))
    assert ns["bitblas"].__version__ == bitblas_amd.__version__ and ns["bitblas"].__backend__ == "bitblas_amd"
    # the type annotation / isinstance check callers make (integration: `from bitblas.ops import Operator`)
    op = ns["Matmul"](ns["MatmulConfig"](M=1, N=256, K=512, A_dtype="float16", W_dtype="int4"), enable_tuning=False)
    assert isinstance(op, ns["Operator"]) and op.is_tilelang_backend() and not op.is_tir_backend()


def test_codegen_subpackages_are_refused_loudly(bitblas_alias):
    for name in ("bitblas.tvm", "bitblas.tl.lower", "bitblas.base.roller", "bitblas.gpu.intrin.lop3", "bitblas.relax"):
        with pytest.raises(ImportError, match="no code generator"):
            __import__(name)
    with pytest.raises(ImportError):          # not a reference module at all: the ordinary error
        __import__("bitblas.no_such_module")


def test_install_is_idempotent_and_refuses_a_foreign_module(monkeypatch):
    assert "bitblas" not in sys.modules
    a = bitblas_amd.install_as_bitblas()
    try:
        assert bitblas_amd.install_as_bitblas() is a is sys.modules["bitblas"]
    finally:
        compat.uninstall()
    foreign = types.ModuleType("bitblas")
    monkeypatch.setitem(sys.modules, "bitblas", foreign)
    with pytest.raises(RuntimeError, match="already imported"):
        bitblas_amd.install_as_bitblas()
    try:
        assert bitblas_amd.install_as_bitblas(force=True).__backend__ == "bitblas_amd"
    finally:
        compat.uninstall()
    assert sys.modules["bitblas"] is foreign          # put back


def test_gen_quant4_roundtrip():
    """the helper the reference's GPTQ tests draw weights from: codes in range, dequantised weight = codes * scale"""
    torch.manual_seed(0)
    w, linear, s, q = bitblas_amd.quantization.gen_quant4(256, 64, 128)
    assert w.shape == (256, 64) and s.shape == (2, 64) and q.shape == (256, 64) and q.dtype == torch.int32
    assert int(q.min()) >= -8 and int(q.max()) <= 8
    want = (q.view(2, 128, 64).half() * s.view(2, 1, 64)).reshape(256, 64).t()
    assert torch.equal(linear.weight.data, want)
    assert float((linear.weight.data.t().float() - w.float()).abs().max()) <= float(s.max()) * 0.5 + 1e-3


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "bitblas", "quantization", "utils.py")), reason="reference checkout not present")
def test_gen_quant4_equals_the_reference_function():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_quant_utils", os.path.join(REF, "bitblas", "quantization", "utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for (k, n, g) in [(256, 64, -1), (256, 64, 128), (512, 256, 32)]:
        torch.manual_seed(7)
        a = ref.gen_quant4(k, n, g)
        torch.manual_seed(7)
        b = bitblas_amd.quantization.gen_quant4(k, n, g)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1].weight.data, b[1].weight.data)
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[3].dtype == b[3].dtype


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "integration", "BitNet", "utils_quant.py")), reason="reference checkout not present")
def test_reference_bitnet_caller_runs_unmodified_on_the_alias(bitblas_alias, tmp_path, capsys):
    """integration/BitNet/utils_quant.py:37-148 executed from its path: `BitLinearBitBLAS` builds its operator through the
    global cache, tunes, saves the database, quantises and packs a float weight - every host-side step of the caller up
    to the launch.  The packed ternary weight it produces equals this package's own BitLinear's byte for byte."""
    from bitblas_amd import cache as wcache
    from bitblas_amd.bitnet import BitLinear
    old_db = wcache.get_database_path()
    wcache.set_database_path(str(tmp_path))
    wcache.global_operator_cache.clear()
    try:
        path = os.path.join(REF, "integration", "BitNet", "utils_quant.py")
        ns = {"__name__": "reference_utils_quant", "__file__": path}
        exec(compile(open(path).read(), path, "exec"), ns)
        assert ns["BITBLAS_DATABASE_PATH"] == str(tmp_path)
        K, N = 512, 256
        torch.manual_seed(0)
        fp = torch.nn.Linear(K, N, bias=False)
        fp.weight.data = torch.randn(N, K) * 0.05
        ref_layer = ns["BitLinearBitBLAS"].from_bit_linear(fp, weight_group=1)
        assert isinstance(ref_layer.bitblas_matmul, bitblas_amd.Matmul)
        assert wcache.global_operator_cache.size() == 1                        # added by the caller after "tuning"
        assert any(f.endswith(".json") for _, _, files in os.walk(tmp_path) for f in files)   # and saved
        # a second layer of the same shape finds the operator in the cache (the caller's own code path)
        again = ns["BitLinearBitBLAS"](K, N)
        assert again.bitblas_matmul is ref_layer.bitblas_matmul
        assert "found in global_operator_cache" in capsys.readouterr().out
        assert tuple(ref_layer.qweight.shape) == (N, K // 4) and ref_layer.qweight.dtype == torch.int8
        mine = BitLinear.from_bit_linear(fp)
        with pytest.raises(NotImplementedError):
            BitLinear.from_bit_linear(fp, weight_group=3)
        assert np.array_equal(ref_layer.qweight.numpy(), mine.qweight.numpy())
        assert torch.allclose(ref_layer.sw.float().reshape(-1)[0], mine.sw.reshape(-1)[0], rtol=0, atol=0)
    finally:
        wcache.set_database_path(old_db)
        wcache.global_operator_cache.clear()
