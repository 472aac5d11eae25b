"""B_decode on its own (`wqaa_dequantize`) and the two-pass member (B_decode to a scratch + the plain GEMM through the
vendor library, `wqaa_matmul_desc.two_pass_min_m`).

* `wqaa_dequantize` against `oracle.dequantize_weight` - the TE graph's first stage, tirscript/matmul_dequantize_impl.py:
  391-449 - BIT FOR BIT for every format / zeros mode / layout: the device decoders pinned on whole matrices (the MFMA
  members use the same routines in their loop);
* the two-pass matmul against the oracle like the fused members (tests/test_gemm_gpu.py), and against the fused member;
* `Matmul.hardware_aware_finetune` measures both on the device and keeps the faster one (the reference's tuner:
  ops/operator.py:262-293).
"""
import ctypes

import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from bitblas_amd import lib as wlib
from helpers import _to_dev, assert_fp_parity, make_case, oracle_output
from test_group_gpu import build

pytestmark = pytest.mark.gpu
DEV = "cuda"

F16_CASES = [
    dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original"),
    dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="rescale"),
    dict(W_dtype="uint4", group_size=64, with_scaling=True, with_zeros=True, zeros_mode="quantized"),
    dict(W_dtype="int4", group_size=128, with_scaling=True),
    dict(W_dtype="int4", group_size=-1, with_scaling=False, fast_decoding=False),
    dict(W_dtype="uint2", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original"),
    dict(W_dtype="int2", group_size=-1, with_scaling=True),
    dict(W_dtype="uint1", group_size=128, with_scaling=True),
    dict(W_dtype="int1", group_size=-1, with_scaling=False),
    dict(W_dtype="uint8", group_size=128, with_scaling=True),
    dict(W_dtype="nf4", group_size=128, with_scaling=True),
    dict(W_dtype="fp4_e2m1", group_size=-1, with_scaling=False),
    dict(W_dtype="e4m3_float8", group_size=128, with_scaling=True),
]


def dequantize_on_device(mm, w):
    W, scale, zeros, _ = w
    cfg = mm.config
    out = torch.empty((cfg.N, cfg.K), dtype=bitblas.matmul.torch_dtype(cfg.A_dtype), device=DEV)
    lut = mm._ensure_lut(torch.device(DEV, torch.cuda.current_device()))
    L = wlib.load_library()
    st = L.wqaa_dequantize(ctypes.byref(mm.lib.desc), W.data_ptr(), lut.data_ptr() if lut is not None else None,
                           scale.data_ptr() if scale is not None else None, zeros.data_ptr() if zeros is not None else None,
                           out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    wlib.check(st)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("kw", F16_CASES, ids=lambda kw: kw["W_dtype"] + "_" + str(kw.get("zeros_mode", "")) + str(kw.get("group_size")))
def test_dequantize_is_the_te_graphs_b_decode_bit_for_bit(kw):
    N, K = 272, 1024
    case = make_case(64, N, K, scale_mul=0.05, seed=len(str(kw)), **dict(kw))
    mm, w = build(case, True)
    got = dequantize_on_device(mm, w).cpu().numpy()
    lut = np.asarray(bitblas.Matmul.NF4_VALUES, dtype=np.float16) if case["source_format"] == "nf" else None
    want = oracle.dequantize_weight(case["codes"], case["source_format"], case["bit"], scale=case["scale"], zeros=case["zeros"],
                                    zeros_mode=case["zeros_mode"], group_size=case["g"], a_dtype="float16", strict_reference=True, lut=lut)
    assert got.dtype == np.float16 and got.shape == (N, K)
    bad = int((got.view(np.uint16) != want.astype(np.float16).view(np.uint16)).sum())
    assert bad == 0, f"{bad} of {got.size} elements differ"


@pytest.mark.parametrize("wd", ["int2", "uint2", "int4", "int1"])
def test_dequantize_int8_operand(wd):
    """int8 activations' operators: B_decode is the integer weight in int8 (BitNet's W_int2 among them)"""
    N, K = 272, 1024
    case = make_case(64, N, K, W_dtype=wd, A_dtype="int8", out_dtype="int32", seed=3)
    mm, w = build(case, True)
    cfg = mm.config
    out = torch.empty((N, K), dtype=torch.int8, device=DEV)
    L = wlib.load_library()
    wlib.check(L.wqaa_dequantize(ctypes.byref(mm.lib.desc), w[0].data_ptr(), None, None, None, out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    want = oracle.dequantize_weight(case["codes"], case["source_format"], case["bit"], a_dtype="int8")
    assert np.array_equal(out.cpu().numpy().astype(np.int64), want)


@pytest.mark.dense_lib
@pytest.mark.parametrize("kw", [F16_CASES[0], F16_CASES[2], F16_CASES[3], F16_CASES[5], F16_CASES[10], F16_CASES[12]],
                         ids=lambda kw: kw["W_dtype"] + "_" + str(kw.get("zeros_mode", "")))
@pytest.mark.parametrize("M", [16, 300, 1024])
def test_two_pass_member_against_the_oracle_and_the_fused_member(kw, M):
    N, K = 528, 1024
    case = make_case(M, N, K, scale_mul=0.05, seed=M + len(str(kw)), out_dtype="float16", **dict(kw))
    mm, w = build(case, True)
    A = _to_dev(case["A"], DEV)
    fused = mm(A, *w)
    assert mm.plans[M]["kernel_family"] == 2
    mm.lib.desc.two_pass_min_m = 16
    plan = mm.lib.plan(M)
    assert plan["kernel_family"] == 4 and plan["name"].endswith("_dq_hipblaslt"), plan
    two = mm(A, *w)
    torch.cuda.synchronize()
    want = oracle_output(case, strict_reference=True)
    assert_fp_parity(two.cpu().numpy(), want, rtol=1e-3, atol_frac=1e-3)
    assert_fp_parity(two.cpu().numpy(), fused.cpu().numpy(), rtol=1e-3, atol_frac=1e-3)
    # below the threshold the fused member stays
    mm.lib.desc.two_pass_min_m = M + 1
    assert mm.lib.plan(M)["kernel_family"] == 2


@pytest.mark.dense_lib
def test_two_pass_int8_is_bit_exact():
    """W_int2 x A_int8 (BASELINE c4) through B_decode (int8) + the library's int8 GEMM: integer results, equal to the oracle"""
    M, N, K = 512, 1024, 2048
    case = make_case(M, N, K, W_dtype="int2", A_dtype="int8", out_dtype="int32", seed=11)
    mm, w = build(case, True)
    mm.lib.desc.two_pass_min_m = 16
    plan = mm.lib.plan(M)
    if plan["kernel_family"] != 4:
        pytest.skip("the vendor library offers no int8 x int8 -> int32 algorithm for this shape")
    out = mm(_to_dev(case["A"], DEV), *w)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle_output(case, strict_reference=True))


@pytest.mark.dense_lib
def test_finetune_measures_and_keeps_the_faster_member():
    cfg = bitblas.MatmulConfig(M=[1, 16, 1024, 4096], N=4096, K=4096, A_dtype="float16", W_dtype="uint4", accum_dtype="float16",
                               out_dtype="float16", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original")
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    assert mm.lib.desc.two_pass_min_m == 0 and mm.plans[4096]["kernel_family"] == 2
    plans = mm.hardware_aware_finetune()
    thr = mm.lib.desc.two_pass_min_m
    tuned = getattr(mm, "_tuned", {})
    if thr:
        assert thr in (1024, 4096) and plans[4096]["kernel_family"] == 4 and plans[16]["kernel_family"] == 2
        assert tuned[thr]["two_pass_ms"] <= 0.97 * tuned[thr]["fused_ms"]
    else:
        assert plans[4096]["kernel_family"] == 2
    # whichever member was kept: same contract against the oracle
    case = make_case(4096, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=1)
    _, w = build(case, True)
    out = mm(_to_dev(case["A"], DEV), *w)
    torch.cuda.synchronize()
    rows = np.random.default_rng(0).choice(4096, size=32, replace=False)
    sub = dict(case, A=case["A"][rows], M=32)
    assert_fp_parity(out[rows].cpu().numpy(), oracle_output(sub), rtol=1e-3, atol_frac=1e-3)


@pytest.mark.dense_lib
@pytest.mark.parametrize("wd,zm", [("uint4", "original"), ("uint4", "quantized"), ("int4", None), ("nf4", None), ("uint2", "rescale")])
def test_two_pass_bfloat16(wd, zm, monkeypatch):
    """bfloat16 activations: B_decode in bfloat16 (one rounding per operation as the TE expression), plain GEMM in the library;
    WQAA_TWO_PASS forces the member at plan time (reference cases: testing/python/operators/test_general_matmul_bf16.py:56-178)"""
    from test_gemm_gpu import _bf16_case
    monkeypatch.setenv("WQAA_TWO_PASS", "16")
    out, want, mm = _bf16_case(300, 528, 1024, wd, 128, True, zm, seed=4)
    assert mm.plans[300]["kernel_family"] == 4, mm.plans[300]
    assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)


@pytest.mark.dense_lib
def test_two_pass_capture_replay_and_two_streams():
    """the two-pass member under the workspace ownership rules: caller-owned scratch shared per (stream, device)
    (`lib.shared_workspace`), capturable into a hipGraph, two streams at once with their own scratch"""
    M, N, K = 512, 4096, 4096              # N * K * 2 B = 32 MiB of B_decode: above the shared-workspace threshold
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=2)
    case2 = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                      scale_mul=0.02, seed=3)
    mm, w = build(case, True)
    _, w2 = build(case2, True)
    mm.lib.desc.two_pass_min_m = 256
    assert mm.lib.plan(M)["kernel_family"] == 4
    assert mm.lib.workspace_bytes(M) >= N * K * 2
    A = _to_dev(case["A"], DEV)
    base = mm(A, *w).clone()
    base2 = mm(A, *w2).clone()
    torch.cuda.synchronize()
    rows = np.arange(0, M, 17)
    assert_fp_parity(base[rows].cpu().numpy(), oracle_output(dict(case, A=case["A"][rows], M=len(rows))), rtol=1e-3, atol_frac=1e-3)
    # capture -> replay with other weights in the same buffers
    wbuf = [t.clone() if t is not None else None for t in w]
    out = torch.empty_like(base)
    mm(A, *wbuf, output=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mm(A, *wbuf, output=out)
    for dst, src in zip(wbuf, w2):
        if dst is not None:
            dst.copy_(src)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, base2)
    # two streams, different weights, many rounds: each stream has its own scratch
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = torch.empty_like(base), torch.empty_like(base)
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            mm(A, *w, output=o1)
        with torch.cuda.stream(s2):
            mm(A, *w2, output=o2)
    torch.cuda.synchronize()
    assert torch.equal(o1, base) and torch.equal(o2, base2)
    # a caller workspace that is too small is refused
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    with pytest.raises(wlib.WqaaError):
        mm.lib.run_ws(A.data_ptr(), w[0].data_ptr(), None, w[1].data_ptr(), w[2].data_ptr(), None, o1.data_ptr(), M,
                      torch.cuda.current_stream().cuda_stream, small.data_ptr(), small.numel())
