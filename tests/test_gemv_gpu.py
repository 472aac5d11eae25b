"""GEMV (M < 8) parity: HIP kernel through the C ABI vs the CPU oracle on the same seeded inputs.

Case list restates the reference's op tests (test_general_matmul_ops_backend_tl.py:327-343 GEMV
half, test_general_matmul_ops_backend.py:211-229, test_general_matmul_ops_nf4.py:64-66,
test_general_matmul_fp8.py:149-158) and adds BASELINE.json configs c1/c2/c4.
Tolerance: 1e-3 relative (north star) with an absolute floor of 1e-3 * rms(output); integer
paths must be bit exact.
"""
import os

import numpy as np
import pytest
import torch

from helpers import set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

# this file covers the per-element-rounding GEMV members (`strict_reference=True`: the reference's definition to the letter);
# the library's default members at M <= 2 - the exact-product family - have tests/test_gemvx_gpu.py, test_group_gpu.py,
# test_float_ops_gpu.py and the reference-held fixtures (test_optest_golden.py, test_te_golden.py, both families)
import functools  # noqa: E402
import helpers as _helpers  # noqa: E402
hip_output = functools.partial(_helpers.hip_output, strict_reference=True)

pytestmark = pytest.mark.gpu

REF_GEMV_CASES = [
    # (M, N, K, W_dtype, group_size, with_scaling, with_zeros, zeros_mode, fast_decoding)
    (1, 256, 256, "uint4", -1, False, False, "original", None),
    (1, 256, 256, "uint4", -1, False, False, "original", False),
    (1, 256, 256, "int4", -1, True, False, "original", None),
    (1, 256, 256, "int4", 32, True, False, "original", None),
    (1, 256, 256, "uint4", 32, True, True, "original", None),
    (1, 256, 256, "uint4", 32, True, True, "rescale", None),
    (1, 256, 256, "uint4", 32, True, True, "quantized", None),
]


@pytest.mark.parametrize("case_args", REF_GEMV_CASES)
def test_reference_gemv_cases(case_args):
    M, N, K, wd, g, ws, wz, zm, fd = case_args
    case = make_case(M, N, K, W_dtype=wd, group_size=g, with_scaling=ws, with_zeros=wz, zeros_mode=zm,
                     fast_decoding=fd)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 1
    assert_fp_parity(got, oracle_output(case))


def test_baseline_c1_plumbing():
    case = make_case(1, 1024, 1024, W_dtype="int4", group_size=-1)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_baseline_c2_llama7b_shapes(N, K):
    case = make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("with_bias", [False, True])
def test_small_batches_and_bias(M, with_bias):
    case = make_case(M, 512, 1024, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                     zeros_mode="original", with_bias=with_bias, seed=M)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("wd", ["uint2", "int2", "uint1", "int1", "uint8", "int8"])
@pytest.mark.parametrize("fd", [None, False])
def test_other_integer_widths_fp16(wd, fd):
    K = 1024
    case = make_case(1, 256, K, W_dtype=wd, group_size=-1, with_scaling=True, fast_decoding=fd,
                     scale_mul=0.05)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


def test_nf4_lut():
    case = make_case(1, 512, 1024, W_dtype="nf4", group_size=128, with_scaling=True)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


def test_fp4_reference_decode():
    case = make_case(2, 256, 512, W_dtype="fp4_e2m1", group_size=-1, with_scaling=True)
    got, _ = hip_output(case)
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("ws,g", [(False, -1), (True, 32)])
def test_e4m3_weight_fp16_activation(strict, ws, g):
    case = make_case(1, 256, 1024, W_dtype="e4m3_float8", group_size=g, with_scaling=ws)
    got, _ = hip_output(case, strict_reference=strict)
    assert_fp_parity(got, oracle_output(case, strict_reference=strict))


@pytest.mark.parametrize("M", [1, 4])
@pytest.mark.parametrize("wd,fd", [("int2", None), ("int2", False), ("int4", None), ("uint4", None), ("int1", None)])
@pytest.mark.parametrize("out_dtype", ["int32", "float32"])
def test_int8_activation_exact(M, wd, fd, out_dtype):
    """BASELINE c4 family (BitNet W_int2 A_int8): int32 accumulation must be bit exact."""
    case = make_case(M, 512, 2048, W_dtype=wd, A_dtype="int8", out_dtype=out_dtype, fast_decoding=fd)
    got, _ = hip_output(case)
    want = oracle_output(case)
    assert np.array_equal(got, want)


def test_baseline_c4_gemv_full_size():
    case = make_case(1, 4096, 4096, W_dtype="int2", A_dtype="int8", out_dtype="int32")
    got, _ = hip_output(case)
    assert np.array_equal(got, oracle_output(case))


def test_dense_fp16_and_int8_gemv():
    rng = np.random.default_rng(0)
    import bitblas_amd as bitblas
    A = (rng.random((1, 1024), dtype=np.float32) - 0.5).astype(np.float16)
    W = (rng.random((256, 1024), dtype=np.float32) - 0.5).astype(np.float16)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=256, K=1024, A_dtype="float16", W_dtype="float16"),
                        enable_tuning=False)
    out = mm(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()).cpu().numpy()
    want = (A.astype(np.float64) @ W.astype(np.float64).T).astype(np.float16)
    assert_fp_parity(out, want)
    A8 = rng.integers(-128, 128, size=(2, 1024), dtype=np.int8)
    W8 = rng.integers(-128, 128, size=(256, 1024), dtype=np.int8)
    mm8 = bitblas.Matmul(bitblas.MatmulConfig(M=2, N=256, K=1024, A_dtype="int8", W_dtype="int8",
                                              accum_dtype="int32", out_dtype="int32"), enable_tuning=False)
    out8 = mm8(torch.from_numpy(A8).cuda(), torch.from_numpy(W8).cuda()).cpu().numpy()
    assert np.array_equal(out8, A8.astype(np.int64) @ W8.astype(np.int64).T)


def test_linearity_property_full_size():
    """Size-independent check at BASELINE c2 size: f(a1 + a2) == f(a1) + f(a2) to fp16 rounding,
    and scaling one weight row's scale scales exactly that output column."""
    case = make_case(1, 4096, 4096, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02)
    got, mm = hip_output(case)
    case2 = dict(case)
    case2["A"] = (case["A"] * np.float16(2.0)).astype(np.float16)
    got2, _ = hip_output(case2, matmul=mm)
    assert_fp_parity(got2, (got.astype(np.float32) * 2).astype(np.float16), rtol=1e-3)
    case3 = dict(case)
    s = case["scale"].copy()
    s[7, :] = 0
    case3["scale"] = s
    got3, _ = hip_output(case3, matmul=mm)
    assert got3[0, 7] == 0
    assert np.array_equal(np.delete(got3, 7, axis=1), np.delete(got, 7, axis=1))


def test_output_buffer_and_stream_semantics():
    case = make_case(1, 256, 256, W_dtype="uint4")
    import bitblas_amd as bitblas
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    W = mm.transform_weight(torch.from_numpy(case["w_user"]).cuda())
    A = torch.from_numpy(case["A"]).cuda()
    out = torch.full((1, 256), 7.0, dtype=torch.float16, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ret = mm(A, W, output=out)
    s.synchronize()
    assert ret.data_ptr() == out.data_ptr()
    assert_fp_parity(out.cpu().numpy(), oracle_output(case))


@pytest.mark.parametrize("wd,g,ws,wz,zm", [("int4", -1, False, False, "original"),
                                            ("uint4", 128, True, True, "original"),
                                            ("uint2", 64, True, True, "quantized"),
                                            ("e4m3_float8", -1, False, False, "original")])
@pytest.mark.parametrize("M", [1, 2])
def test_register_and_lds_activation_members_agree(wd, g, ws, wz, zm, M, monkeypatch):
    """M <= 2, K within one step runs the member that keeps its activation slices in registers; the
    LDS-staged member (forced through the selector's tuning switch) must give the same bits."""
    case = make_case(M, 512, 2048 if wd != "uint2" else 4096, W_dtype=wd, group_size=g, with_scaling=ws,
                     with_zeros=wz, zeros_mode=zm)
    got, mm = hip_output(case)
    # (2-bit weights x float16 at M = 2: 128 dwords of activations per lane do not fit the register file - that member spilled
    # and is no longer built, csrc/wqaa_gemv_kernel.h gemv_direct_fits; the LDS-staged member is the only one)
    fits = not (wd == "uint2" and M == 2)
    assert mm.plans[M]["name"].endswith("_areg") == fits, mm.plans[M]["name"]
    set_knobs(monkeypatch, "gemv", areg=0)
    got_lds, mm2 = hip_output(case)
    assert not mm2.plans[M]["name"].endswith("_areg")
    assert np.array_equal(got, got_lds)
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("wd,ad", [("uint1", "float16"), ("int1", "float16"), ("uint2", "float16"), ("int1", "int8")])
@pytest.mark.parametrize("M", [1, 2])
def test_lds_staged_low_bit_members_agree_with_the_register_members(wd, ad, M, monkeypatch):
    """1-bit (any M) and 2-bit (M = 2) weights run the LDS-staged members (their register-resident twins spilled and are gone:
    round 3, profiles/r03_ab_direct_fit.txt); where a register-resident member still exists (2-bit M = 1, int1 x int8 M = 1)
    the two must give the same bits, and every case must match the oracle."""
    int8 = ad == "int8"
    case = make_case(M, 512, 4096, W_dtype=wd, A_dtype=ad, out_dtype="int32" if int8 else "float16",
                     **({} if int8 else dict(group_size=128, with_scaling=True, scale_mul=0.05)))
    got, mm = hip_output(case)
    set_knobs(monkeypatch, "gemv", areg=0)
    got_lds, mm2 = hip_output(case)
    assert not mm2.plans[M]["name"].endswith("_areg")
    assert np.array_equal(got, got_lds), (mm.plans[M]["name"], mm2.plans[M]["name"])
    if int8:
        assert np.array_equal(got, oracle_output(case))
    else:
        assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("wd,N,K", [("int2", 512, 4096), ("uint2", 264, 4096), ("int1", 512, 8192), ("int2", 300, 3200)])
@pytest.mark.parametrize("M", [1, 2])
def test_integer_gemv_one_chunk_rows_agree_with_the_two_chunk_members(wd, N, K, M, monkeypatch):
    """Rows of ONE lane chunk (2-bit weights at K = 4096: 64 lanes x 64 weights; K = 3200 fills 50 lanes of it) run the
    (4 rows x 1 chunk) integer-activation members (`..._b?r4d1`); `WQAA_GEMV_TUNE=chunk=0` pins the (2 x 2) members that pad the row with
    a second chunk: same int32 bits, both equal to the oracle."""
    case = make_case(M, N, K, W_dtype=wd, A_dtype="int8", out_dtype="int32", seed=N + K + M)
    got, mm = hip_output(case)
    assert "r4d1" in mm.plans[M]["name"], mm.plans[M]["name"]
    set_knobs(monkeypatch, "gemv", chunk="0")
    got2, mm2 = hip_output(case)
    assert "d2" in mm2.plans[M]["name"] and "r4d1" not in mm2.plans[M]["name"], mm2.plans[M]["name"]
    assert np.array_equal(got, got2)
    assert np.array_equal(got, oracle_output(case))


def test_non_contiguous_activations_are_read_correctly():
    """A strided view (every other row of a taller matrix, and a 3-d batch) must not be read as raw memory."""
    import bitblas_amd as bitblas
    case = make_case(4, 256, 512, W_dtype="int4")
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=[1, 4, 16], N=256, K=512, A_dtype="float16", W_dtype="int4"), enable_tuning=False)
    W = mm.transform_weight(torch.from_numpy(case["w_user"]).cuda())
    tall = torch.zeros((8, 512), dtype=torch.float16, device="cuda")
    tall[0::2] = torch.from_numpy(case["A"]).cuda()
    tall[1::2] = 123.0
    out = mm(tall[0::2], W)
    assert_fp_parity(out.cpu().numpy(), oracle_output(case))
    out3 = mm(tall[0::2].reshape(2, 2, 512), W)
    assert out3.shape == (2, 2, 256)
    assert_fp_parity(out3.reshape(4, 256).cpu().numpy(), oracle_output(case))
    with pytest.raises(ValueError):
        mm(tall[0::2], W, output=torch.empty((4, 512), dtype=torch.float16, device="cuda")[:, ::2])


# ---- K split across the waves of a workgroup (rounding members; few-row shards of a column-parallel layer) ----
@pytest.mark.parametrize("kw", [0, 2, 3, 4, 7])
@pytest.mark.parametrize("N,K", [(1024, 28672), (1280, 8192), (512, 11008), (100, 8192)])
def test_k_split_rounding_members(N, K, kw, monkeypatch):
    """kw = 0: the selector's own choice (must split for these shapes); otherwise forced.  Same parity bound as the
    unsplit member, bit-identical from run to run (the parts meet in LDS in a fixed order)."""
    if kw:
        set_knobs(monkeypatch, "gemv", kw=str(kw))
    case = make_case(1, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=kw + N, out_dtype="float32", accum_dtype="float32")
    got, mm = hip_output(case)
    plan = mm.plans[1]
    assert "_gemvx_" not in plan["name"] and plan["split_k"] > 1, plan
    if kw:
        assert plan["split_k"] == min(kw, plan["split_k"]) and plan["name"].endswith(f"k{plan['split_k']}"), plan
    assert_fp_parity(got, oracle_output(case))
    got2, _ = hip_output(case, matmul=mm)
    assert np.array_equal(got.view(np.uint32), got2.view(np.uint32))


@pytest.mark.parametrize("M", [1, 2])                # the K-split twins exist for the M <= 2 tiles (M >= 3 is MFMA territory)
@pytest.mark.parametrize("zm", [None, "rescale", "quantized"])
def test_k_split_batches_zero_modes_and_bias(M, zm, monkeypatch):
    set_knobs(monkeypatch, "gemv", kw="4")
    case = make_case(M, 384 + 2, 16384, W_dtype="uint4" if zm else "int4", group_size=128, with_scaling=True, with_zeros=zm is not None,
                     zeros_mode=zm or "original", with_bias=True, scale_mul=0.02, seed=M)
    got, mm = hip_output(case)
    assert mm.plans[M]["split_k"] == 4, mm.plans[M]
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("wd", ["int2", "int4"])
def test_k_split_int8_activations_bit_exact(wd, monkeypatch):
    set_knobs(monkeypatch, "gemv", kw="3")
    case = make_case(2, 768, 24576, W_dtype=wd, A_dtype="int8", out_dtype="int32", seed=3)
    got, mm = hip_output(case)
    assert mm.plans[2]["split_k"] == 3, mm.plans[2]
    assert np.array_equal(got, oracle_output(case))


def test_k_split_request_from_the_operator():
    import bitblas_amd as bitblas
    cfg = bitblas.MatmulConfigWithSplitK(M=1, N=1024, K=16384, A_dtype="float16", W_dtype="int4", group_size=128, with_scaling=True, k_split=2)
    mm = bitblas.MatmulWithSplitK(cfg, enable_tuning=False, strict_reference=True)          # the rounding member
    assert mm.plans[1]["split_k"] == 2 and "_gemvx_" not in mm.plans[1]["name"], mm.plans[1]
