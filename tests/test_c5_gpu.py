"""BASELINE config c5 at full size: W_fp8(e4m3) x A_fp8(e4m3) on the Llama-3-70B linear shapes (hidden 8192,
intermediate 28672, 64 q / 8 kv heads), M = 4096 (MFMA GEMM) and M = 1 (GEMV), unsharded and as the per-rank
column shards N/8 of the multi-GPU design.

Reference: the dense TE definition `C = A @ W^T`, fp32 accumulate (bitblas/ops/general_matmul/tirscript/
matmul_impl.py:50-86) and the fp8 operator test (testing/python/operators/test_general_matmul_fp8.py:11-71, which
builds the same operands - `torch.rand(...).to(float8_e4m3fn)` - prints its expectation and asserts nothing).
Parity here is against the oracle's exact OCP decode + fp64 matmul on a sample of output rows x columns (every
output element depends on one activation row and one weight row only), tolerance 1e-4 relative + 1e-4 * rms
(fp32 summation order inside the matrix core), fp16 outputs within one fp16 rounding of that.

These are the shapes `bench.py` times under `members` (gemm_fp8_*): the ping-pong 256x256 member built on
v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/wqaa_gemm_pp_kernel.h, plan suffix `pp`) must be the one that runs them with
float16 output, and it must be correct; float32 output stays on the lockstep 256x256x256 member.
"""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from helpers import assert_fp_parity

import bitblas_amd as bitblas

pytestmark = pytest.mark.gpu

TDT = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}

# (name, N, K) of one unsharded GPU; gate_up = the fused gate|up projection of SURVEY.md section 8(d)
LLAMA3_70B = [("o_proj", 8192, 8192), ("down_proj", 8192, 28672), ("qkv_proj", 10240, 8192),
              ("gate_proj", 28672, 8192), ("gate_up_proj", 57344, 8192)]


def _operands(M, N, K, a_dt, w_dt, seed):
    """uniform [-1, 1) operands generated on the device (a 470 MB weight matrix is slow to build on the host)"""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    A = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).to(TDT[a_dt])
    W = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).to(TDT[w_dt])
    return A, W


def _sample(n, count, edges, rng):
    """`count` indices of range(n): tile edges first (0, last, around multiples of 256), the rest random"""
    fixed = [i for i in edges if 0 <= i < n]
    rest = rng.choice(n, size=max(0, min(n, count) - len(fixed)), replace=False).tolist() if n > len(fixed) else []
    return np.unique(np.array(fixed + rest, dtype=np.int64))


def _check(M, N, K, a_dt="e4m3_float8", w_dt="e4m3_float8", out_dtype="float16", want_plan=None, seed=0,
           n_rows=64, n_cols=1024):
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=a_dt, W_dtype=w_dt, accum_dtype="float32",
                                             out_dtype=out_dtype), enable_tuning=False)
    if want_plan is not None:
        assert want_plan in mm.plans[M]["name"], mm.plans[M]["name"]
    A, W = _operands(M, N, K, a_dt, w_dt, seed)
    out = mm(A, W)
    torch.cuda.synchronize()
    rng = np.random.default_rng(seed)
    rows = _sample(M, n_rows, [0, 1, 15, 16, 255, 256, M - 1], rng)
    cols = _sample(N, n_cols, [0, 1, 15, 16, 255, 256, 257, N - 257, N - 256, N - 1], rng)
    ri, ci = torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()
    got = out[ri][:, ci].float().cpu().numpy()
    want = oracle.matmul_dense(A[ri].view(torch.int8).cpu().numpy(), W[ci].view(torch.int8).cpu().numpy(),
                               a_dtype=a_dt, w_dtype=w_dt, out_dtype="float32")
    assert np.isfinite(got).all()
    if out_dtype == "float16":
        # one fp16 rounding of an fp32 sum: 2^-11 relative, plus the summation-order term
        assert_fp_parity(got, want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-4)
        # and no systematic error: the mean signed deviation is far below one fp16 ulp of the rms
        rms = float(np.sqrt(np.mean(want ** 2)))
        assert abs(float(np.mean(got - want))) < 2e-4 * rms
    else:
        assert_fp_parity(got, want, rtol=1e-4, atol_frac=1e-4)
    return mm


@pytest.mark.parametrize("name,N,K", LLAMA3_70B)
def test_c5_gemm_m4096_unsharded(name, N, K):
    """the shapes bench.py times: every one must run the ping-pong 256x256 member"""
    _check(4096, N, K, want_plan="tcx256x256x128pp", seed=N // 256 + K // 1024)


@pytest.mark.parametrize("name,N,K", [("o_proj", 8192, 8192), ("down_proj", 8192, 28672)])
def test_c5_gemm_m4096_every_output_element(name, N, K):
    """whole M x N outputs, no sampling: the oracle's exact OCP decode, its product through a threaded float64 GEMM
    (oracle/wqaa_oracle.py: _gemm_nt) - seconds even for the 1.9 TFLOP down projection"""
    _check(4096, N, K, want_plan="tcx256x256x128pp", seed=N // 128 + K // 512, n_rows=4096, n_cols=N)


@pytest.mark.parametrize("name,N,K", LLAMA3_70B)
def test_c5_gemm_m4096_per_rank_shard(name, N, K):
    """column shard of 8-way tensor parallelism: N' = N / 8; fp32 output checked exactly"""
    _check(4096, N // 8, K, out_dtype="float32", seed=N // 8 + 1)


@pytest.mark.parametrize("name,N,K", LLAMA3_70B)
@pytest.mark.parametrize("shard", [1, 8])
def test_c5_gemv_m1(name, N, K, shard):
    mm = _check(1, N // shard, K, out_dtype="float32", seed=N + shard, n_rows=1, n_cols=4096)
    assert mm.plans[1]["kernel_family"] == 1


@pytest.mark.parametrize("a_dt,w_dt", [("e5m2_float8", "e5m2_float8"), ("e4m3_float8", "e5m2_float8"),
                                       ("e5m2_float8", "e4m3_float8")])
def test_c5_other_fp8_pairings_full_size(a_dt, w_dt):
    """the mixed pairs of `is_native_compute` (general_matmul/__init__.py:33-51) on the o_proj shape"""
    _check(4096, 8192, 8192, a_dt=a_dt, w_dt=w_dt, want_plan="tcx256x256x128pp", seed=5)


def test_c5_ragged_m_on_the_large_member():
    """M not a multiple of the 256-row tile: the tail rows of the last tile are never stored, the rest are exact"""
    _check(4096 - 37, 8192, 8192, out_dtype="float32", seed=7)


def test_c5_row_permutation_property():
    """size-independent property at full size: permuting activation rows permutes output rows, bit for bit"""
    M, N, K = 4096, 8192, 8192
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8",
                                             accum_dtype="float32", out_dtype="float16"), enable_tuning=False)
    A, W = _operands(M, N, K, "e4m3_float8", "e4m3_float8", 11)
    perm = torch.randperm(M, device="cuda")
    out = mm(A, W)
    out_p = mm(A.view(torch.int8)[perm].view(TDT["e4m3_float8"]), W)
    torch.cuda.synchronize()
    assert torch.equal(out[perm].view(torch.int16), out_p.view(torch.int16))
