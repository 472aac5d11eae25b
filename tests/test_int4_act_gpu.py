"""W_int4 / W_int2 x A_int4 (BitNet a4.8 family): packed int4 activations, int32 accumulation, bit exact.

Restates the non-propagated cases of the reference's test_general_matmul_ops_int4.py:149-155 (which
draws non-negative operands only) and widens them to the full nibble range, GEMV and ragged M.
"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle

pytestmark = pytest.mark.gpu


def pack_nibbles(x):
    """(rows, K) small ints -> (rows, K/2) int8, low nibble = even k (the reference test's packing, :49-50)"""
    u = x.astype(np.int64) & 0xF
    return (u[:, 0::2] | (u[:, 1::2] << 4)).astype(np.uint8).view(np.int8)


def run_case(M, N, K, W_dtype, fast_decoding, lo_a=-8, hi_a=8, seed=0, out_dtype="int32"):
    rng = np.random.default_rng(seed)
    A = rng.integers(lo_a, hi_a, size=(M, K))
    bits = 4 if W_dtype == "int4" else 2
    if bits == 4:
        W = rng.integers(-8, 8, size=(N, K))            # native two's-complement nibbles
    else:
        W = rng.integers(0, 4, size=(N, K))             # raw 2-bit fields, zero-extended by the kernels
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int4", W_dtype=W_dtype, accum_dtype="int32",
                               out_dtype=out_dtype, layout="nt", propagate_b=False, fast_decoding=fast_decoding)
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    A_packed = pack_nibbles(A)
    codes = (W & ((1 << bits) - 1)).astype(np.int8)
    if bits == 4:
        Wdev = torch.from_numpy(pack_nibbles(W)).cuda()                     # what the reference test feeds (:50, :68)
        assert torch.equal(mm.transform_weight(torch.from_numpy(codes)).cpu(), Wdev.cpu())
    else:
        Wdev = mm.weight_transform(torch.from_numpy(codes)).cuda()          # compress (+ LOP3 interleave)
    out = mm(torch.from_numpy(A_packed).cuda(), Wdev)
    torch.cuda.synchronize()
    want = oracle.matmul_int4_act(A_packed, codes, w_bits=bits, out_dtype=out_dtype)
    direct = (A @ W.T)
    assert np.array_equal(want.astype(np.int64), direct)                     # the oracle against plain integers
    assert np.array_equal(out.cpu().numpy(), want)
    return mm


@pytest.mark.parametrize("W_dtype,fast_decoding", [("int4", False), ("int2", False), ("int2", True)])
def test_reference_int4_cases(W_dtype, fast_decoding):
    """128 x 128 x 128, operands in [0, 4) / [0, 2) as the reference draws them (:44-46, :70)"""
    mm = run_case(128, 128, 128, W_dtype, fast_decoding, lo_a=0, hi_a=4)
    assert mm.plans[128]["kernel_family"] == 1    # K = 128 is below the 256-deep int8 MFMA k-step: batch-tiled GEMV family
    mm = run_case(128, 128, 512, W_dtype, fast_decoding, lo_a=0, hi_a=4)
    assert mm.plans[128]["kernel_family"] == 2


@pytest.mark.parametrize("M", [1, 3, 16, 100, 256, 300])
@pytest.mark.parametrize("W_dtype,fast_decoding", [("int4", False), ("int2", False), ("int2", True), ("int2", None)])
def test_full_nibble_range_all_m(M, W_dtype, fast_decoding):
    run_case(M, 256, 1024, W_dtype, fast_decoding, seed=M)


def test_float32_output_and_llm_shape():
    run_case(1, 4096, 4096, "int2", None, out_dtype="float32")
    run_case(64, 4096, 4096, "int4", False)
