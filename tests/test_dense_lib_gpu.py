"""The plain dense GEMMs through hipBLASLt (csrc/wqaa_dense_lib.hip): W_dtype == A_dtype (e4m3 / e5m2 / float16 / bfloat16),
no scale / zeros / bias, M >= 16 - BASELINE c5 at M = 4096 and the float16 fallbacks (reference semantics: C = A . W^T with
fp32 accumulation, bitblas/ops/general_matmul/tirscript/matmul_impl.py:50-86; reference test:
testing/python/operators/test_general_matmul_fp8.py:11-71).  Checked against the oracle like the own members are
(tests/test_c5_gpu.py), against the own members themselves, and for the ownership rules of the workspace."""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from helpers import assert_fp_parity

pytestmark = [pytest.mark.gpu, pytest.mark.dense_lib]
DEV = "cuda"
TORCH = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2, "float16": torch.float16, "bfloat16": torch.bfloat16}


def operands(M, N, K, dt, seed):
    g = torch.Generator().manual_seed(seed)
    if dt.endswith("float8"):
        A = (torch.rand((M, K), generator=g) * 2 - 1).to(TORCH[dt])
        W = (torch.rand((N, K), generator=g) * 2 - 1).to(TORCH[dt])
    else:
        A = (torch.rand((M, K), generator=g) - 0.5).to(TORCH[dt])
        W = (torch.rand((N, K), generator=g) - 0.5).to(TORCH[dt])
    return A, W


def op_for(M, N, K, dt, out_dtype="float16"):
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=dt, W_dtype=dt, accum_dtype="float32", out_dtype=out_dtype)
    return bitblas.Matmul(cfg, enable_tuning=False)


def reference_rows(A, W, rows, out_dtype):
    """fp32 matmul of the exactly decoded operands on sampled rows (what oracle.matmul_dense computes), cast to out_dtype"""
    want = A[rows].float().double() @ W.float().double().T
    return want.float().to(out_dtype).float().numpy()


@pytest.mark.parametrize("name,N,K", [("o", 8192, 8192), ("down", 8192, 28672), ("qkv", 10240, 8192), ("gate", 28672, 8192),
                                      ("o_shard", 1024, 8192), ("qkv_shard", 1280, 8192)])
def test_c5_m4096_on_the_vendor_library(name, N, K):
    M = 4096
    A, W = operands(M, N, K, "e4m3_float8", N // 256 + K // 1024)
    op = op_for(M, N, K, "e4m3_float8")
    assert op.plans[M]["kernel_family"] == 3 and op.plans[M]["name"].endswith("_hipblaslt"), op.plans[M]
    out = op(A.to(DEV), W.to(DEV))
    torch.cuda.synchronize()
    rows = np.random.default_rng(N + K).choice(M, size=64, replace=False)
    assert_fp_parity(out[rows].float().cpu().numpy(), reference_rows(A, W, rows, torch.float16), rtol=1e-3, atol_frac=1e-3)


@pytest.mark.parametrize("dt", ["e4m3_float8", "e5m2_float8", "float16", "bfloat16"])
@pytest.mark.parametrize("M", [16, 100, 512])
@pytest.mark.parametrize("out_dtype", ["float16", "float32"])
def test_dtypes_and_row_counts_against_the_oracle_and_the_own_members(dt, M, out_dtype, monkeypatch):
    if dt == "bfloat16" and out_dtype == "float16":
        out_dtype = "bfloat16"
    N, K = 272, 1024
    A, W = operands(M, N, K, dt, M + len(dt))
    op = op_for(M, N, K, dt, out_dtype)
    assert op.plans[M]["kernel_family"] == 3, op.plans[M]
    out = op(A.to(DEV), W.to(DEV))
    torch.cuda.synchronize()
    want = reference_rows(A, W, np.arange(M), TORCH.get(out_dtype, torch.float32))
    assert_fp_parity(out.float().cpu().numpy(), want, rtol=1e-3 if out_dtype != "bfloat16" else 8e-3, atol_frac=1e-3 if out_dtype != "bfloat16" else 8e-3)
    # the own member on the same operands (WQAA_DENSE_LIB=0 is a plan-time switch)
    monkeypatch.setenv("WQAA_DENSE_LIB", "0")
    own = op_for(M, N, K, dt, out_dtype)                 # planned under the switch
    assert own.plans[M]["kernel_family"] == 2, own.plans[M]
    out2 = own(A.to(DEV), W.to(DEV))
    torch.cuda.synchronize()
    assert_fp_parity(out.float().cpu().numpy(), out2.float().cpu().numpy(), rtol=1e-3 if out_dtype != "bfloat16" else 8e-3,
                     atol_frac=1e-3 if out_dtype != "bfloat16" else 8e-3)


def test_what_stays_on_the_own_members():
    """M < 16, a bias, integer pairs and quantised weights never go to the vendor library"""
    assert op_for(8, 1024, 1024, "float16").plans[8]["kernel_family"] in (1, 2)
    cfg = bitblas.MatmulConfig(M=256, N=1024, K=1024, A_dtype="float16", W_dtype="float16", accum_dtype="float32", out_dtype="float16",
                               with_bias=True)
    assert bitblas.Matmul(cfg, enable_tuning=False).plans[256]["kernel_family"] == 2
    cfg = bitblas.MatmulConfig(M=256, N=1024, K=1024, A_dtype="int8", W_dtype="int8", accum_dtype="int32", out_dtype="int32")
    assert bitblas.Matmul(cfg, enable_tuning=False).plans[256]["kernel_family"] == 2
    cfg = bitblas.MatmulConfig(M=256, N=1024, K=1024, A_dtype="float16", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")
    assert bitblas.Matmul(cfg, enable_tuning=False).plans[256]["kernel_family"] == 2


def test_capture_replay_two_streams_and_caller_workspace():
    """the library path under the ownership rules of the split-K scratch (tests/test_workspace_gpu.py): capturable, two
    streams at once, caller-owned workspace through wqaa_matmul_opts"""
    M, N, K = 512, 2048, 4096
    A, W = operands(M, N, K, "e4m3_float8", 3)
    A2, _ = operands(M, N, K, "e4m3_float8", 4)
    Ad, A2d, Wd = A.to(DEV), A2.to(DEV), W.to(DEV)
    op = op_for(M, N, K, "e4m3_float8")
    base = op(Ad, Wd).clone()
    base2 = op(A2d, Wd).clone()
    torch.cuda.synchronize()
    # capture + replay on new activations
    xs = Ad.clone()
    out = torch.empty_like(base)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        op(xs, Wd, output=out)
    xs.copy_(A2d)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, base2)
    # two streams running the same operator concurrently
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = torch.empty_like(base), torch.empty_like(base)
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(s1):
            op(Ad, Wd, output=o1)
        with torch.cuda.stream(s2):
            op(A2d, Wd, output=o2)
    torch.cuda.synchronize()
    assert torch.equal(o1, base) and torch.equal(o2, base2)
    # caller-owned workspace: the need is reported, a short one is refused
    need = op.lib.workspace_bytes(M)
    if need:
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        o3 = torch.empty_like(base)
        op.lib.run_ws(Ad.data_ptr(), Wd.data_ptr(), None, None, None, None, o3.data_ptr(), M, torch.cuda.current_stream().cuda_stream,
                      ws.data_ptr(), need)
        torch.cuda.synchronize()
        assert torch.equal(o3, base)
        with pytest.raises(bitblas.lib.WqaaError):
            op.lib.run_ws(Ad.data_ptr(), Wd.data_ptr(), None, None, None, None, o3.data_ptr(), M, torch.cuda.current_stream().cuda_stream,
                          ws.data_ptr(), 16)


def test_measured_tuning_keeps_results_and_never_a_slower_algorithm():
    """`Matmul.hardware_aware_finetune` -> `wqaa_tune`: the library's candidate algorithms are timed on the device; whichever
    is kept, the result still meets the contract, and the operator is not slower than before (3 % margin + noise)"""
    M, N, K = 2048, 10240, 8192                     # the c5 shape on which the heuristic's first choice is a slow one
    A, W = operands(M, N, K, "e4m3_float8", 8)
    Ad, Wd = A.to(DEV), W.to(DEV)
    op = op_for(M, N, K, "e4m3_float8")

    def ms(n=8, repeats=3):
        # best of `repeats` bursts: a fresh box ramps its clocks, and one slow burst must not fail the comparison
        out = torch.empty((M, N), dtype=torch.float16, device=DEV)
        for _ in range(3):
            op(Ad, Wd, output=out)
        best = float("inf")
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                op(Ad, Wd, output=out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        return best, out

    before, ref = ms()
    op.hardware_aware_finetune()
    # (the vendor library's tuned algorithm - family 3 - or, since the partial-round tail of round 4 made this library's own
    # member the faster one on this shape, the own member: the tuner keeps whichever it timed fastest)
    assert op.plans[M]["kernel_family"] in (2, 3)
    after, out = ms()
    assert after <= 1.25 * before, (before, after)
    rows = np.arange(0, M, 37)
    assert_fp_parity(out[rows].float().cpu().numpy(), reference_rows(A, W, rows, torch.float16), rtol=1e-3, atol_frac=1e-3)
    assert_fp_parity(out[rows].float().cpu().numpy(), ref[rows].float().cpu().numpy(), rtol=1e-3, atol_frac=1e-3)


def test_tuning_may_hand_a_shape_back_to_the_own_member():
    """`wqaa_tune` also times this library's own MFMA member: where the vendor heuristic has a hole (seen: e4m3 M = 256 at
    8192 x 28672) the shape goes back to the own member; either way the tuned operator is correct and not slower"""
    M, N, K = 256, 8192, 28672
    A, W = operands(M, N, K, "e4m3_float8", 21)
    Ad, Wd = A.to(DEV), W.to(DEV)
    op = op_for(M, N, K, "e4m3_float8")
    assert op.plans[M]["kernel_family"] == 3

    def ms(n=8, repeats=3):
        # best of `repeats` bursts: a fresh box ramps its clocks, and one slow burst must not fail the comparison
        out = torch.empty((M, N), dtype=torch.float16, device=DEV)
        for _ in range(3):
            op(Ad, Wd, output=out)
        best = float("inf")
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                op(Ad, Wd, output=out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        return best, out

    before, _ = ms()
    op.hardware_aware_finetune()
    after, out = ms()
    assert op.plans[M]["kernel_family"] in (2, 3)
    assert after <= 1.25 * before, (before, after, op.plans[M])
    rows = np.arange(0, M, 9)
    assert_fp_parity(out[rows].float().cpu().numpy(), reference_rows(A, W, rows, torch.float16), rtol=1e-3, atol_frac=1e-3)
    print("M=256 8192x28672 e4m3: before", before, "after", after, op.plans[M]["name"])
