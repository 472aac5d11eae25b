"""Golden matmul vectors produced by RUNNING the reference's own operator tests.

`import bitblas` is impossible here (its tvm / tilelang submodules are empty), but the reference's op tests
build their seeded inputs and their expected result (`ref_program`, a plain-torch formulation) around the
`bitblas.Matmul` call.  This script executes those test functions unmodified, from the files where they lie
under /root/reference, with a *recorder* standing in for the `bitblas` package:

  * `bitblas.Matmul(...)` is a recorder object: it keeps the config, hands out the reference's own
    `BITBLAS_TRICK_DTYPE_MAP` and NF4 table (both parsed out of the reference source with `ast`),
    an identity `weight_transform` (the integer codes pass through unpacked) and records the operands of the
    call; it computes nothing;
  * `bitblas.testing.torch_assert_close` / `torch.testing.assert_close` record (actual, expected) instead of
    comparing - `expected` is the reference test's `ref_result`;
  * `bitblas.quantization.general_compress` is the reference's real function (numpy-only file, loaded
    standalone); `Tensor.cuda()` is the identity (no GPU here).

Every recorded case = the reference test's seeded operands + the reference test's own expected output.
Large cases are cut to their first rows of A / first columns of the output (C[:r, :c] depends only on A[:r] and
W[:c]) so the fixtures stay small.  Output: tests/golden/optest_golden.npz + optest_golden.json (committed).

Runs only where /root/reference exists.  Test infrastructure - never imported by the product.
Sources executed: testing/python/operators/test_general_matmul_ops_backend_tl.py (:327-343, 13 cases),
test_general_matmul_ops_backend.py (:211-229, 9 cases, the ones with bias),
test_general_matmul_fp8.py (:150-158; :63-71 dense e4m3 / e5m2 - that test PRINTS its expectation and asserts
nothing, the printed tensor is recorded), test_general_matmul_ops_nf4.py (:64-66), test_general_matmul_bf16.py
(:170-178); testing/python/module/test_bitblas_linear.py (:45-49 dense fp16 Linear vs torch.nn.Linear, :169-176
uint4 / uint2 weight-only, with bias) through a recorder `bitblas.Linear`.
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OPTESTS = os.path.join(REF, "testing", "python", "operators")
GM_INIT = os.path.join(REF, "bitblas", "ops", "general_matmul", "__init__.py")
REF_UTILS = os.path.join(REF, "bitblas", "quantization", "utils.py")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_NPZ = os.path.join(HERE, "..", "tests", "golden", "optest_golden.npz")
OUT_JSON = os.path.join(HERE, "..", "tests", "golden", "optest_golden.json")
MAX_ROWS, MAX_COLS = 64, 128


def parse_reference_constants():
    """BITBLAS_TRICK_DTYPE_MAP (general_matmul/__init__.py:323-344) and the NF4 table (:413-434) as literals."""
    tree = ast.parse(open(GM_INIT).read())
    trick, lut = None, None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id == "BITBLAS_TRICK_DTYPE_MAP" for t in node.targets):
            trick = ast.literal_eval(node.value)
        if isinstance(node, ast.List) and len(node.elts) == 16 and lut is None:
            try:
                vals = ast.literal_eval(node)
            except ValueError:
                continue
            if vals[0] == -1.0 and vals[-1] == 1.0 and vals[7] == 0.0:
                lut = vals
    assert trick is not None and lut is not None
    return trick, lut


def load_reference_utils():
    spec = importlib.util.spec_from_file_location("ref_quant_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Recorder:
    def __init__(self):
        self.cases = []
        self.current = None

    def begin(self, config):
        self.current = {"config": config, "operands": None, "expected": None}

    def operands(self, args, output, placeholder):
        self.current["operands"] = [a for a in args]
        self.current["outputs"] = (output, placeholder)

    def expected(self, actual, expected):
        # the reference tests pass (output, ref) or (ref, output): the expectation is whichever tensor is NOT the
        # output buffer handed to / returned by the recorder (an nf4 test's `torch.empty` output holds garbage)
        outs = self.current.pop("outputs")
        if any(actual is o for o in outs if o is not None):
            exp = expected
        elif any(expected is o for o in outs if o is not None):
            exp = actual
        else:
            raise AssertionError("neither side of the comparison is the recorder's output")
        self.current["expected"] = exp
        self.cases.append(self.current)
        self.current = None


def make_print_hook(rec):
    """test_general_matmul_fp8.py:11-60 prints `torch_ref_out` / `bitblas_out` and asserts nothing: the printed
    expectation is recorded as the case's expected value."""
    state = {}

    def hook(*args, **kw):
        if args and args[0] == "torch_ref_out" and rec.current is not None:
            state["exp"] = args[1]
        elif args and args[0] == "bitblas_out" and rec.current is not None and "exp" in state:
            rec.current.pop("outputs", None)
            rec.current["expected"] = state.pop("exp")
            rec.cases.append(rec.current)
            rec.current = None
    return hook


def install_stub(rec, trick, lut, ref_utils):
    import torch

    class MatmulConfig:
        def __init__(self, **kw):
            self.kw = dict(kw)
            for k, v in kw.items():
                setattr(self, k, v)

    class Matmul:
        BITBLAS_TRICK_DTYPE_MAP = trick

        def __init__(self, config, enable_tuning=False, backend=None, **_):
            self.config = config
            rec.begin(dict(config.kw))
            self.weight_transform = lambda w: w        # the codes pass through unpacked
            self.source_format, self.bit = trick[config.kw["W_dtype"]]
            a_dtype = config.kw["A_dtype"]
            self.lut = torch.tensor(lut, dtype=getattr(torch, a_dtype)) if self.source_format == "nf" else None
            self.scheduled_ir_module = "<recorder>"

        def transform_weight(self, w, *_, **__):
            return w

        def get_source(self):
            return ""

        def __call__(self, *args, output=None):
            M = args[0].shape[0]
            out_dtype = self.config.kw["out_dtype"]
            tmap = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}
            placeholder = torch.zeros((M, self.config.kw["N"]), dtype=tmap.get(out_dtype) or getattr(torch, out_dtype))
            rec.operands(args, output, placeholder)
            return placeholder

    class Linear(torch.nn.Module):
        """recorder for `bitblas.Linear` (testing/python/module/test_bitblas_linear.py): holds the buffers the test
        assigns, records them with the input at call time, computes nothing"""

        def __init__(self, in_features, out_features, bias=False, A_dtype="float16", W_dtype="float16",
                     accum_dtype="float16", out_dtype="float16", group_size=-1, with_scaling=False,
                     with_zeros=False, zeros_mode=None, opt_M=None, **_):
            super().__init__()
            self.kw = dict(N=out_features, K=in_features, A_dtype=A_dtype, W_dtype=W_dtype, accum_dtype=accum_dtype,
                           out_dtype=out_dtype, group_size=group_size, with_scaling=with_scaling,
                           with_zeros=with_zeros, zeros_mode=zeros_mode, with_bias=bool(bias))
            src, bit = trick[W_dtype]
            self.bitblas_matmul = types.SimpleNamespace(source_format=src, bit=bit, weight_transform=lambda w: w)
            holder = lambda: torch.nn.Parameter(torch.zeros(1), requires_grad=False)   # `.data = ...` targets
            self.qweight, self.scales, self.zeros = holder(), holder(), holder()
            self.bias = holder() if bias else None
            self.weight = None

        def cuda(self, *a, **k):
            return self

        def load_and_transform_weight(self, weight, *a, **k):
            self.weight = weight

        def forward(self, x):
            kw = dict(self.kw, M=x.shape[0])
            rec.begin(kw)
            dense = self.weight is not None
            ops = [x, self.weight if dense else self.qweight.data]
            if kw["with_scaling"]:
                ops.append(self.scales.data)
            if kw["with_zeros"]:
                ops.append(self.zeros.data)
            if kw["with_bias"]:
                ops.append(self.bias.data)
            placeholder = torch.zeros((x.shape[0], kw["N"]), dtype=torch.float16)
            rec.operands(ops, None, placeholder)
            return placeholder

    bb = types.ModuleType("bitblas")
    bb.MatmulConfig, bb.Matmul, bb.Linear = MatmulConfig, Matmul, Linear
    bb.set_log_level = lambda *a, **k: None
    testing = types.ModuleType("bitblas.testing")
    testing.torch_assert_close = lambda a, b, **k: rec.expected(a, b)
    testing.requires_cuda_compute_version = lambda *a, **k: (lambda f: f)
    testing.main = lambda: None
    quant = types.ModuleType("bitblas.quantization")
    quant.general_compress = ref_utils.general_compress
    bb.testing, bb.quantization = testing, quant
    bb.__path__ = []                                    # a package, so `from bitblas.tl.lower import ...` resolves
    cache = types.ModuleType("bitblas.cache")
    cache.global_operator_cache = types.SimpleNamespace(clear=lambda: None)
    qutils = types.ModuleType("bitblas.quantization.utils")
    qutils.general_compress = ref_utils.general_compress
    quant.__path__ = []
    quant.utils = qutils
    bb.cache = cache
    sys.modules.update({"bitblas.cache": cache, "bitblas.quantization.utils": qutils})
    tl = types.ModuleType("bitblas.tl")
    tl.__path__ = []
    lower = types.ModuleType("bitblas.tl.lower")
    lower.tl_lower = None                                # imported by a test module, used by its codegen tests only
    tl.lower = lower
    bb.tl = tl
    sys.modules.update({"bitblas": bb, "bitblas.testing": testing, "bitblas.quantization": quant,
                        "bitblas.tl": tl, "bitblas.tl.lower": lower})
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.testing.assert_close = lambda a, b, **k: rec.expected(a, b)


def run_reference_test(filename, func, print_hook=None):
    path = os.path.join(OPTESTS if not filename.startswith("module/") else os.path.join(REF, "testing", "python"), filename)
    ns = {"__name__": "ref_optest", "__file__": path}
    if print_hook is not None:
        ns["print"] = print_hook                         # module-global shadow of the builtin, this module only
    exec(compile(open(path).read(), path, "exec"), ns)   # the reference's test module, as it lies
    ns[func]()


def to_numpy(t):
    import torch
    if t is None:
        return None
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().copy(), "bfloat16"
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.int8).numpy().copy(), str(t.dtype).replace("torch.", "")
    return t.numpy().copy(), str(t.dtype).replace("torch.", "")


def main():
    if not os.path.exists(REF_UTILS):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    import torch
    trick, lut = parse_reference_constants()
    ref_utils = load_reference_utils()
    rec = Recorder()
    install_stub(rec, trick, lut, ref_utils)
    plan = [("test_general_matmul_ops_backend_tl.py", "test_matmul_torch_dequant_forward"),
            ("test_general_matmul_ops_backend.py", "test_matmul_torch_forward"),
            ("test_general_matmul_fp8.py", "test_matmul_torch_forward_weight_dequantize"),
            ("test_general_matmul_fp8.py", "test_matmul_torch_forward"),
            ("test_general_matmul_ops_nf4.py", "test_matmul_torch_forward"),
            ("test_general_matmul_bf16.py", "test_matmul_torch_forward_weight_dequantize"),
            ("module/test_bitblas_linear.py", "test_correctness_consistent"),
            ("module/test_bitblas_linear.py", "test_correctness_weight_only_dequantize")]
    origin = []
    for fn, func in plan:
        n0 = len(rec.cases)
        run_reference_test(fn, func, make_print_hook(rec) if func == "test_matmul_torch_forward" and "fp8" in fn else None)
        origin += [fn] * (len(rec.cases) - n0)
        print(f"{fn}::{func}: {len(rec.cases) - n0} cases", file=sys.stderr)

    arrays, meta = {}, []
    for i, (case, src) in enumerate(zip(rec.cases, origin)):
        cfg = {k: v for k, v in case["config"].items() if v is not None}
        M, N, K = cfg["M"], cfg["N"], cfg["K"]
        r, c = min(M, MAX_ROWS), min(N, MAX_COLS if not src.startswith("module/") else 32)
        ops = list(case["operands"])
        names = ["A", "W"]
        if cfg.get("with_scaling"):
            names.append("scale")
        if cfg.get("with_zeros"):
            names.append("zeros")
        if cfg.get("with_bias"):
            names.append("bias")
        assert len(ops) == len(names), (cfg, len(ops), names)
        entry = {"source": src, "config": cfg, "rows": r, "cols": c, "dtypes": {}}
        for name, t in zip(names, ops):
            if name == "A":
                t = t[:r]
            elif name in ("W", "scale", "bias"):
                t = t[:c]
            elif name == "zeros":
                if cfg.get("zeros_mode") == "quantized":
                    _, bit = trick[cfg["W_dtype"]]
                    t = t[:, :c * bit // 8]
                else:
                    t = t[:c]
            arr, dt = to_numpy(t.contiguous())
            arrays[f"c{i}_{name}"] = arr
            entry["dtypes"][name] = dt
        exp, dt = to_numpy(case["expected"][:r, :c].contiguous())
        arrays[f"c{i}_expected"] = exp
        entry["dtypes"]["expected"] = dt
        meta.append(entry)
    np.savez_compressed(OUT_NPZ, **arrays)
    with open(OUT_JSON, "w") as f:
        json.dump({"generator": "oracle/gen_optest_golden.py", "torch": torch.__version__, "cases": meta}, f, indent=1)
    print(f"wrote {len(meta)} cases, {os.path.getsize(OUT_NPZ) / 1e6:.2f} MB", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
