"""`Linear.enable_decoded_weight_cache` on the device (bitblas_amd/module.py): a resident B_decode + the plain dense GEMM
must give what the packed-weight members give - the oracle's contract - and follow the live buffers.
(Named to run last: new at the end of round 2, written without a GPU at hand.)"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from helpers import assert_fp_parity

pytestmark = pytest.mark.gpu


def make(N, K, wd, zm, bias, opt_M):
    rng = np.random.default_rng(N + K)
    bits = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[wd][1]
    lin = bitblas.Linear(K, N, bias=bias, A_dtype="float16", W_dtype=wd, group_size=128, with_scaling=True,
                         with_zeros=zm is not None, zeros_mode=zm, opt_M=opt_M, enable_tuning=False)
    codes = rng.integers(0, 1 << bits, size=(N, K)).astype(np.int8)
    scale = (rng.random((N, K // 128)) * 0.05).astype(np.float16)
    zeros = None
    if zm == "original":
        zeros = np.full((N, K // 128), float(1 << (bits - 1)), dtype=np.float16)
    b = rng.standard_normal(N).astype(np.float16) if bias else None
    lin.load_and_transform_weight(torch.from_numpy(codes), scales=torch.from_numpy(scale),
                                  zeros=None if zeros is None else torch.from_numpy(zeros),
                                  bias=None if b is None else torch.from_numpy(b))
    return lin.cuda(), codes, scale, zeros, b, bits


def want_rows(A, codes, bits, scale, zeros, zm, b, rows):
    return oracle.matmul_dequant(A[rows], codes, source_format="uint", bit=bits, scale=scale, zeros=zeros,
                                 zeros_mode=zm or "original", group_size=128, bias=b)


@pytest.mark.parametrize("wd,zm,bias", [("uint4", "original", False), ("uint4", None, True), ("uint2", "original", False)])
def test_decoded_cache_matches_the_packed_members_and_the_oracle(wd, zm, bias):
    N, K, M = 512, 1024, 300
    lin, codes, scale, zeros, b, bits = make(N, K, wd, zm, bias, [1, 16, 512])
    rng = np.random.default_rng(5)
    A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
    Ad = torch.from_numpy(A).cuda()
    packed = lin(Ad).cpu().numpy()
    lin.enable_decoded_weight_cache(min_m=64)
    cached = lin(Ad).cpu().numpy()
    assert lin._decoded is not None and tuple(lin._decoded.shape) == (N, K)
    rows = np.arange(0, M, 7)
    assert_fp_parity(cached[rows], want_rows(A, codes, bits, scale, zeros, zm, b, rows), rtol=1e-3, atol_frac=1e-3)
    assert_fp_parity(cached, packed, rtol=2e-3, atol_frac=2e-3)      # two members, each within 1e-3 of the oracle
    # the resident copy IS the TE graph's B_decode: bit-identical to the oracle's
    d = oracle.dequantize_weight(codes, "uint", bits, K=K, scale=scale, zeros=zeros, zeros_mode=zm or "original", group_size=128)
    assert np.array_equal(lin._decoded.cpu().numpy().view(np.uint16), np.asarray(d, dtype=np.float16).view(np.uint16))
    small = lin(Ad[:8]).cpu().numpy()            # below the threshold: the packed path, untouched
    assert_fp_parity(small, packed[:8], rtol=2e-3, atol_frac=2e-3)      # M = 8 and M = 300 take different members


def test_decoded_cache_follows_in_place_updates():
    N, K, M = 256, 512, 128
    lin, codes, scale, zeros, b, bits = make(N, K, "uint4", "original", False, [1, 16, 256])
    lin.enable_decoded_weight_cache(min_m=64)
    A = torch.from_numpy((np.random.default_rng(2).random((M, K), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    out1 = lin(A).clone()
    ptr = lin._decoded.data_ptr()
    lin.scales.mul_(2)                           # in place: same pointers, new version
    out2 = lin(A)
    torch.cuda.synchronize()
    assert lin._decoded.data_ptr() == ptr        # re-decoded into the same resident buffer
    assert_fp_parity(out2.cpu().numpy(), (out1.float() * 2).half().cpu().numpy(), rtol=2e-3, atol_frac=2e-3)
    lin.disable_decoded_weight_cache()
    out3 = lin(A)
    assert_fp_parity(out3.cpu().numpy(), out2.cpu().numpy(), rtol=2e-3, atol_frac=2e-3)
