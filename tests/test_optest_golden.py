"""The reference's own operator tests as fixtures: seeded operands and the expected result of their in-test
`ref_program`, produced by RUNNING those test functions (oracle/gen_optest_golden.py; sources
testing/python/operators/test_general_matmul_ops_backend_tl.py:327-343, test_general_matmul_ops_backend.py:211-229,
test_general_matmul_fp8.py:150-158,
test_general_matmul_ops_nf4.py:64-66, test_general_matmul_bf16.py:170-178, module/test_bitblas_linear.py:45-49, 169-176).

CPU: pins the oracle's decode + matmul semantics against those expectations.
GPU: the HIP path through the C ABI on the same operands against the same expectations."""
import json
import os

import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
import bitblas_amd as bitblas
from helpers import assert_fp_parity

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "optest_golden.json")))["cases"]
ARR = np.load(os.path.join(GOLD, "optest_golden.npz"))
IDS = [f"{i}-{c['source'].replace('test_general_matmul_', '').replace('.py', '').replace('module/test_bitblas_', '')}-{c['config']['W_dtype']}-m{c['rows']}"
       f"{'-g%d' % c['config']['group_size'] if c['config'].get('group_size') else ''}"
       f"{'-' + (c['config'].get('zeros_mode') or 'original') if c['config'].get('with_zeros') else ''}" for i, c in enumerate(META)]


def _bf16_bits_to_f32(a):
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def load(i):
    c = META[i]
    cfg = c["config"]
    out = dict(cfg=cfg, rows=c["rows"], cols=c["cols"])
    for name, dt in c["dtypes"].items():
        a = ARR[f"c{i}_{name}"]
        out[name] = _bf16_bits_to_f32(a) if dt == "bfloat16" else a
        out[name + "_dt"] = dt
    out["K"] = cfg["K"]
    g = cfg.get("group_size")
    out["g"] = cfg["K"] if g in (None, -1) else g
    out["src"], out["bit"] = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[cfg["W_dtype"]]
    out["zeros_mode"] = cfg.get("zeros_mode") or "original"
    return out


def tolerance(c, slack=1.0):
    # bfloat16 expectations are a bf16 x bf16 torch.matmul: the RESULT is rounded to bfloat16 (2^-8 relative).
    # fp16 cases: the oracle is bit-identical to the expectation on >= 99.8 % of the elements, one fp16 ulp off on
    # the rest (fp32 summation order inside torch.matmul)
    t = 8e-3 if c["cfg"]["A_dtype"] == "bfloat16" else 1e-3
    return dict(rtol=t * slack, atol_frac=t * slack)


def test_fixture_inventory():
    srcs = [c["source"] for c in META]
    assert srcs.count("test_general_matmul_ops_backend_tl.py") == 13
    assert srcs.count("test_general_matmul_ops_backend.py") == 9
    assert srcs.count("test_general_matmul_fp8.py") == 8      # 4 weight-dequantize + 4 dense (printed, unasserted upstream)
    assert len(GPU_CASES) == 32
    assert srcs.count("test_general_matmul_ops_nf4.py") == 2
    assert srcs.count("test_general_matmul_bf16.py") == 4
    assert srcs.count("module/test_bitblas_linear.py") == 11


@pytest.mark.parametrize("i", range(len(META)), ids=IDS)
def test_oracle_reproduces_the_reference_tests_expectation(i):
    c = load(i)
    cfg = c["cfg"]
    if cfg["W_dtype"] == "float16":
        # dense fp16 Linear against torch.nn.Linear (module/test_bitblas_linear.py:14-49)
        want = oracle.matmul_dense(c["A"], c["W"], a_dtype="float16", out_dtype="float16", bias=c.get("bias"))
        assert_fp_parity(want, c["expected"], **tolerance(c))
        return
    if cfg["A_dtype"].endswith("float8"):
        # dense fp8 x fp8 (test_general_matmul_fp8.py:11-71): exact products of exactly decoded operands, fp32 sum
        want = oracle.matmul_dense(c["A"], c["W"], a_dtype=cfg["A_dtype"], w_dtype=cfg["W_dtype"], out_dtype=cfg["out_dtype"])
        assert_fp_parity(want, c["expected"], rtol=1e-5, atol_frac=1e-5)
        return
    # the fp8 test's expectation decodes e4m3 per IEEE (`torch_b.to(float16)`), not with the kernels' bit trick
    # (quantization.py:169-176: zero -> 2^-7, subnormals wrong; ~1.5 % of uniform(-1,1) weights are subnormal)
    want = oracle.matmul_dequant(
        c["A"], c["W"], source_format=c["src"], bit=c["bit"], scale=c.get("scale"), zeros=c.get("zeros"),
        zeros_mode=c["zeros_mode"], group_size=c["g"], bias=c.get("bias"), a_dtype=cfg["A_dtype"],
        out_dtype=cfg["out_dtype"], strict_reference=c["src"] != "fp_e4m3")
    assert_fp_parity(want, c["expected"], **tolerance(c))


# dense fp8 x fp8 cases pin the oracle only (the HIP dense members are compared with the oracle in test_gemm_gpu.py)
# (so do the Linear-test fixtures: Linear itself is compared with the oracle in test_linear_gpu.py)
GPU_CASES = [i for i, c in enumerate(META)
             if c["config"]["A_dtype"] in ("float16", "bfloat16") and not c["source"].startswith("module/")]


@pytest.mark.gpu
@pytest.mark.parametrize("members", ["default", "strict_reference"])
@pytest.mark.parametrize("i", GPU_CASES, ids=[IDS[i] for i in GPU_CASES])
def test_hip_path_reproduces_the_reference_tests_expectation(i, members):
    """members = default: the operator exactly as a caller of the reference constructs it (no extra argument: at M <= 2 the
    exact-product GEMV family); strict_reference: the per-element-rounding members.  Both against the expectation the
    reference's own test computed; the achieved error goes to $WQAA_PARITY_MARGINS (profiles/r04_parity_margins.txt)."""
    c = load(i)
    cfg = dict(c["cfg"])
    cfg.update(M=c["rows"], N=c["cols"])
    cfg.pop("propagate_a", None)
    cfg.pop("propagate_b", None)
    config = bitblas.MatmulConfig(**cfg)
    if members == "default":
        mm = bitblas.Matmul(config, enable_tuning=False)
    else:
        mm = bitblas.Matmul(config, enable_tuning=False, strict_reference=c["src"] != "fp_e4m3")
    tdt = {"bfloat16": torch.bfloat16, "float16": torch.float16}[cfg["A_dtype"]]
    A = torch.from_numpy(np.ascontiguousarray(c["A"])).to(tdt).cuda()
    W = torch.from_numpy(np.ascontiguousarray(c["W"]))
    if c["src"] == "fp_e4m3":
        W = mm.transform_weight(W.view(torch.float8_e4m3fn).cuda())
    else:   # integer codes / table indices, as the reference tests hand them to weight_transform
        W = mm.weight_transform(W).cuda() if mm.weight_transform is not None else W.cuda()
    scale = zeros = bias = None
    if "bias" in c:
        bias = torch.from_numpy(np.ascontiguousarray(c["bias"])).to(tdt).cuda()
    if "scale" in c:
        scale = torch.from_numpy(np.ascontiguousarray(c["scale"])).to(tdt).cuda()
    if "zeros" in c:
        z = torch.from_numpy(np.ascontiguousarray(c["zeros"]))
        zeros = z.cuda() if c["zeros_mode"] == "quantized" else z.to(tdt).cuda()
    out = mm(A, W, scale=scale, zeros=zeros, bias=bias)
    torch.cuda.synchronize()
    # two summation orders stack here (kernel vs oracle vs torch): twice the oracle's bound - still 5x tighter than
    # the reference test's own rtol = atol = 1e-2 with 5 % mismatches allowed (backend_tl.py:275)
    from helpers import record_margin
    record_margin(f"optest/{IDS[i]}/{members}/{mm.plans[c['rows']]['name']}", out.float().cpu().numpy(), c["expected"])
    from helpers import contract
    assert_fp_parity(out.float().cpu().numpy(), c["expected"],
                     **contract(cfg["K"], default_members=members == "default", m=c["rows"], bf16=cfg["A_dtype"] == "bfloat16",
                                group_size=cfg.get("group_size"), zeros_mode=c.get("zeros_mode")))
