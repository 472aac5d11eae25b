"""The exact-product GEMV members (`strict_reference=False`, csrc/wqaa_gemvx_kernel.h): sub-byte integer weights x
float16 activations at M <= 2, the dequantised weight never rounded to float16 (denormal-field dot products, zero
point through the chunk's activation sum, group scale on the fp32 partial sum), K optionally split across the waves
of a workgroup.

Checked against
  * the oracle's UNROUNDED formulation (`matmul_dequant_exact`) at fp32-accumulation tolerance - this is what the
    members compute;
  * the reference's definition (`matmul_dequant`, per-element float16 rounding of B_decode - the TE graph executed in
    tests/test_te_golden.py) within the north star's 1e-3 contract: the two differ by the float16 rounding the exact
    members skip, ~2e-4 of the output rms;
  * and against the real-valued result the exact members must be at least as close as the strict ones.
Reference operator tests restated: testing/python/operators/test_general_matmul_ops_backend_tl.py:327-343 (M = 1 rows).
"""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from helpers import set_knobs, case_contract, contract, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def exact_members_wherever_they_exist(request, monkeypatch):
    """The selector hands some shapes (M = 2 on few rows, very long K) to the rounding members because those are faster
    there; the parity tests below are about the exact members themselves, so they lift those fences (WQAA_GEMV_TUNE=exact=2).
    Tests of the selector's own choice opt out with `@pytest.mark.selector_choice`."""
    if request.node.get_closest_marker("selector_choice") is None:
        set_knobs(monkeypatch, "gemv", exact="2")


def exact_output(case):
    return oracle.matmul_dequant_exact(case["A"], case["codes"], source_format=case["source_format"], bit=case["bit"],
                                       scale=case["scale"], zeros=case["zeros"], zeros_mode=case["zeros_mode"],
                                       group_size=case["g"], bias=case["bias"], out_dtype=case["out_dtype"])


def check(case, plan_kw=None):
    got, mm = hip_output(case, strict_reference=False)
    M = case["M"]
    name = mm.plans[M]["name"]
    assert "_gemvx_" in name, name
    if plan_kw is not None:
        assert name.endswith(f"k{plan_kw}"), name
    want_exact = exact_output(case)
    if case["out_dtype"] == "float16":
        # one float16 rounding of the fp32 sum
        assert_fp_parity(got, want_exact, rtol=1e-3, atol_frac=6e-4)    # + the float16 bias add
    else:
        assert_fp_parity(got, want_exact, rtol=2e-5, atol_frac=2e-5)
    assert_fp_parity(got, oracle_output(case), **case_contract(case, default_members=True, m=M))      # the reference's definition: include/wqaa.h's contract
    return got, mm


REF_M1 = [  # (N, K, W_dtype, group, scaling, zeros, zeros_mode) - the M = 1 rows of the reference's list
    (1024, 1024, "uint4", -1, False, False, "original"),
    (1024, 1024, "int4", -1, False, False, "original"),
    (1024, 1024, "int4", -1, True, False, "original"),
    (1024, 1024, "int4", 128, True, False, "original"),
    (1024, 1024, "uint4", 128, True, True, "original"),
    (1024, 1024, "uint4", 128, True, True, "rescale"),
    (1024, 1024, "uint4", 128, True, True, "quantized"),
]


@pytest.mark.parametrize("fd", [None, False])
@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("args", REF_M1)
def test_reference_m1_cases_exact_members(args, M, fd):
    N, K, wd, g, ws, wz, zm = args
    case = make_case(M, N, K, W_dtype=wd, group_size=g, with_scaling=ws, with_zeros=wz, zeros_mode=zm, fast_decoding=fd,
                     scale_mul=0.05, seed=M + K, out_dtype="float32", accum_dtype="float32")
    check(case)


@pytest.mark.parametrize("wd", ["uint2", "int2", "uint1", "int1"])
@pytest.mark.parametrize("fd", [None, False])
@pytest.mark.parametrize("zm", [None, "original", "quantized"])
def test_two_and_one_bit_weights(wd, fd, zm):
    if zm is not None and wd.startswith("int"):
        pytest.skip("zero points pair with unsigned formats (general_matmul/__init__.py:382-385)")
    case = make_case(1, 512, 2048, W_dtype=wd, group_size=128, with_scaling=True, with_zeros=zm is not None, zeros_mode=zm or "original",
                     fast_decoding=fd, scale_mul=0.05, seed=3, out_dtype="float32", accum_dtype="float32")
    check(case)


@pytest.mark.selector_choice
def test_two_rows_on_few_rows_or_long_k_keeps_the_rounding_member():
    """M = 2 with K > 8192 on enough rows to fill the chip without a K split, or on fewer than 8192 rows: the rounding
    member is the faster one there (csrc/wqaa_gemvx.hip: gemvx_eligible), and either numerics meets the contract"""
    for M, N, K in ((2, 4096, 11008), (2, 4096, 4096)):
        case = make_case(M, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=5)
        got, mm = hip_output(case, strict_reference=False)
        assert "_gemvx_" not in mm.plans[M]["name"], mm.plans[M]["name"]
        assert_fp_parity(got, oracle_output(case))
        strict, _ = hip_output(case, strict_reference=True)
        assert np.array_equal(got, strict)


@pytest.mark.selector_choice
@pytest.mark.parametrize("N,K,threads", [(4096, 14336, 1024), (4096, 11008, 1024), (5120, 13824, 512), (1024, 28672, 896)])
def test_long_k_one_row_batch_takes_sixteen_wave_workgroups_where_they_fill_the_chip(N, K, threads):
    """M = 1, three lane-chunk steps or more: the exact members, with twice the rows per workgroup (the activation row is
    staged once per workgroup) wherever that still fills the chip in whole rounds - 5120 x 13824 would leave 320
    workgroups for 256 CUs and keeps 8 waves (csrc/wqaa_gemvx.hip: gemvx_choose)"""
    case = make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=N // 64)
    got, mm = check(case)
    assert mm.plans[1]["threads"] == threads, mm.plans[1]


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (1024, 11008)])
def test_baseline_c2_full_size(N, K):
    """BASELINE c2: W_int4 A_fp16 GEMV, M = 1, Llama-2-7B linear shapes, g = 128 - what bench.py times"""
    case = make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=N // 128)
    got, mm = check(case)
    # closer to the real-valued product than the strict members (which round B_decode to float16 per element)
    strict, _ = hip_output(case, strict_reference=True)
    real = oracle.matmul_dequant_exact(case["A"], case["codes"], source_format="int", bit=4, scale=case["scale"], group_size=128,
                                       out_dtype="float32")
    err_exact = float(np.sqrt(np.mean((got.astype(np.float64) - real) ** 2)))
    err_strict = float(np.sqrt(np.mean((strict.astype(np.float64) - real) ** 2)))
    assert err_exact <= err_strict * 1.05


@pytest.mark.parametrize("kw", [1, 2, 3, 4, 7])
@pytest.mark.parametrize("N,K", [(1024, 28672), (1280, 8192), (512, 11008), (100, 4096)])
def test_k_split_across_the_waves_of_a_workgroup(N, K, kw, monkeypatch):
    """per-rank shards of the multi-GPU split (N / 8 rows x long K) and ragged shapes, every K-split width: the parts
    meet in LDS in a fixed order, so the result is bit-identical run to run; kw > steps is clipped"""
    set_knobs(monkeypatch, "gemv", kw=str(kw))
    case = make_case(1, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=kw + N, out_dtype="float32", accum_dtype="float32")
    got, mm = check(case)
    got2, _ = hip_output(case, strict_reference=False, matmul=mm)
    assert np.array_equal(got.view(np.uint32), got2.view(np.uint32))


@pytest.mark.selector_choice
def test_selector_splits_k_for_few_rows():
    import bitblas_amd as bitblas
    for (N, K) in ((1024, 28672), (1280, 8192)):
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype="int4", group_size=128, with_scaling=True),
                            enable_tuning=False, strict_reference=False)
        assert mm.plans[1]["split_k"] > 1, mm.plans[1]


@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("out_dtype", ["float16", "float32"])
def test_ragged_n_bias_and_outputs(with_bias, out_dtype):
    case = make_case(2, 1000 + 1, 2048 + 64, W_dtype="int4", group_size=64, with_scaling=True, with_bias=with_bias, scale_mul=0.05,
                     seed=9, out_dtype=out_dtype, accum_dtype="float32")
    check(case)


@pytest.mark.selector_choice
def test_strict_reference_keeps_the_rounding_members():
    case = make_case(1, 1024, 1024, W_dtype="int4", group_size=128, with_scaling=True, seed=1)
    _, mm = hip_output(case, strict_reference=True)
    assert "_gemvx_" not in mm.plans[1]["name"]


@pytest.mark.parametrize("areg", [0, 1])
@pytest.mark.parametrize("N,K,zm", [(4096, 4096, None), (11008, 4096, "original"), (1000, 2048 + 64, "rescale"), (516, 4096, "quantized")])
def test_register_resident_activations_member(N, K, zm, areg, monkeypatch):
    """4-bit LOP3 weights, M = 1, K within one step: the lane keeps its own activations in registers (no LDS tile, no
    barrier); forced on and off it must meet the same bounds, ragged K / N and every zero mode included"""
    set_knobs(monkeypatch, "gemv", areg=str(areg))
    case = make_case(1, N, K, W_dtype="uint4" if zm else "int4", group_size=64 if K % 128 else 128, with_scaling=True,
                     with_zeros=zm is not None, zeros_mode=zm or "original", scale_mul=0.02, seed=N + areg,
                     out_dtype="float32", accum_dtype="float32")
    got, mm = check(case)
    assert mm.plans[1]["name"].endswith("_areg") == bool(areg), mm.plans[1]["name"]
