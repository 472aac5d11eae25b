"""The shapes BASELINE.json's `metric` is quoted on: W_int4 x A_fp16 at M = 1 and M = 4096 on the Llama-70B linears - the
reference's own benchmark table, benchmark/README.md:60-62 (V10-V12: 8192 x 8192, 28672 x 8192, 8192 x 28672 at M = 1) and
:73-75 (M10-M12, the same at large M) - plus the 10240 x 8192 q/k/v width of grouped-query attention (64 q + 8 + 8 kv heads).
These are the members bench.py reports as `gemv_int4_n*k*` / `gemm_uint4_m4096_n*k*` / `group_int4_70b_*`.

Every element of every output is compared with the oracle (whole M x N at M = 4096: the oracle decodes the 235 M element
matrices in threaded row blocks and multiplies through a threaded GEMM - seconds, oracle/wqaa_oracle.py), plus the
size-independent properties the domain offers: linearity in A at M = 1, bit-identity of a group launch with its single
calls, identical activation rows -> identical output rows at M = 4096."""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from helpers import _to_dev, assert_fp_parity, case_contract, hip_output, make_case, oracle_output

import bitblas_amd as bitblas

pytestmark = pytest.mark.gpu

LLAMA70B = [(8192, 8192), (28672, 8192), (8192, 28672), (10240, 8192)]


@pytest.mark.parametrize("N,K", LLAMA70B)
def test_gemv_int4_m1_llama70b_default_and_strict_members(N, K):
    """M = 1, int4 g128 + scale (what bench.py times): the default (exact-product) member within the contract of
    include/wqaa.h against the TE definition and at fp32-accumulation tolerance against the unrounded product; the
    strict_reference member at 1e-3 + 1e-3"""
    case = make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=N // 128 + K // 1024)
    got, mm = hip_output(case)
    assert mm.plans[1]["kernel_family"] == 1 and "_gemvx_" in mm.plans[1]["name"], mm.plans[1]
    want = oracle_output(case)
    assert_fp_parity(got, want, **case_contract(case, default_members=True, m=1))
    real = oracle.matmul_dequant_exact(case["A"], case["codes"], source_format="int", bit=4, scale=case["scale"], group_size=128,
                                       out_dtype="float32")
    assert_fp_parity(got, real.astype(np.float16), rtol=1e-3, atol_frac=6e-4)
    strict, mms = hip_output(case, strict_reference=True)
    assert "_gemv_" in mms.plans[1]["name"], mms.plans[1]
    assert_fp_parity(strict, want)
    # the exact-product member is at least as close to the real-valued product as the per-element-rounding one
    err_exact = float(np.sqrt(np.mean((got.astype(np.float64) - real) ** 2)))
    err_strict = float(np.sqrt(np.mean((strict.astype(np.float64) - real) ** 2)))
    assert err_exact <= err_strict * 1.05


@pytest.mark.parametrize("N,K", [(28672, 8192), (8192, 28672)])
def test_gemv_int4_m1_llama70b_linearity(N, K):
    """size-independent property: the operator is linear in A - f(a) + f(b) == f(a + b) to fp16 rounding of the three outputs"""
    case = make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=7)
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    sc = _to_dev(case["scale"], "cuda")
    rng = np.random.default_rng(5)
    # operands on a coarse grid so that a + b is exact in float16
    a = torch.from_numpy((rng.integers(-64, 64, size=(1, K)) / 256.0).astype(np.float16)).cuda()
    b = torch.from_numpy((rng.integers(-64, 64, size=(1, K)) / 256.0).astype(np.float16)).cuda()
    fa, fb, fab = mm(a, W, scale=sc).float(), mm(b, W, scale=sc).float(), mm(a + b, W, scale=sc).float()
    torch.cuda.synchronize()
    rms = float(fab.pow(2).mean().sqrt())
    assert float((fa + fb - fab).abs().max()) <= 3e-3 * rms + 2e-3 * float(fab.abs().max())


@pytest.mark.parametrize("Ns,K", [((8192, 1024, 1024), 8192), ((28672, 28672), 8192)])
def test_group_launch_llama70b_bit_identical_to_single_calls(Ns, K):
    """q/k/v (grouped-query attention: 8192 + 1024 + 1024 rows) and gate/up (2 x 28672) of a 70B layer as ONE launch each
    (wqaa_matmul_group): the bits of the single calls, and those meet the oracle"""
    cases = [make_case(1, N, K, W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.02, seed=N + i) for i, N in enumerate(Ns)]
    ops = [bitblas.Matmul(c["config"], enable_tuning=False) for c in cases]
    Ws = [(op.weight_transform(torch.from_numpy(c["codes"])).cuda(), _to_dev(c["scale"], "cuda")) for op, c in zip(ops, cases)]
    A = _to_dev(cases[0]["A"], "cuda")
    plan = bitblas.group_plan(ops, 1)
    outs = bitblas.matmul_group(ops, A, Ws)
    torch.cuda.synchronize()
    for op, c, w, o in zip(ops, cases, Ws, outs):
        assert torch.equal(o, op(A, *w)), plan
        c["A"] = cases[0]["A"]
        assert_fp_parity(o.cpu().numpy(), oracle_output(c), **case_contract(c, default_members=True, m=1))


@pytest.mark.parametrize("N,K", LLAMA70B[:3])
def test_gemm_uint4_m4096_llama70b_every_output_element(N, K):
    """M = 4096, uint4 g128 + scale + zeros (BASELINE c3's format on the 70B shapes): the whole 4096 x N output against the
    oracle; identical activation rows give identical output rows"""
    case = make_case(4096, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=N // 256 + K // 512)
    got, mm = hip_output(case)
    assert mm.plans[4096]["kernel_family"] == 2 and mm.plans[4096]["name"].endswith("pp"), mm.plans[4096]
    want = oracle.matmul_dequant(case["A"], case["codes"], source_format="uint", bit=4, scale=case["scale"], zeros=case["zeros"],
                                 zeros_mode="original", group_size=128, wide=False)
    assert_fp_parity(got, want)
    del want
    case2 = dict(case)
    A2 = case["A"].copy()
    A2[1::2] = A2[0::2]
    case2["A"] = A2
    got2, _ = hip_output(case2, matmul=mm)
    assert np.array_equal(got2[1::2], got2[0::2])
    assert np.array_equal(got2[0::2], got[0::2])
