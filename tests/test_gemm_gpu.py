"""MFMA GEMM (M >= 8) parity: HIP kernel through the C ABI vs the CPU oracle on the same seeded inputs.

Case list restates the GEMM half of the reference's op tests
(testing/python/operators/test_general_matmul_ops_backend_tl.py:336-343: M=256, N=K=256, uint4/int4,
group -1/32, the three zero modes) and adds BASELINE.json configs c3 (uint4 g=128 + zeros,
M in {16,128,4096}, N=K=4096) and c4 (int2 x int8, bit exact).  Tolerance: 1e-3 relative with an
absolute floor of 1e-3 * rms(output); integer paths are bit exact.
"""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from helpers import set_knobs, case_contract, contract, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu

REF_GEMM_CASES = [
    # (M, N, K, W_dtype, group_size, with_scaling, with_zeros, zeros_mode, fast_decoding)
    (256, 256, 256, "uint4", -1, False, False, "original", None),
    (256, 256, 256, "uint4", -1, False, False, "original", False),
    (256, 256, 256, "int4", -1, True, False, "original", None),
    (256, 256, 256, "int4", 32, True, False, "original", None),
    (256, 256, 256, "uint4", 32, True, True, "original", None),
    (256, 256, 256, "uint4", 32, True, True, "rescale", None),
    (256, 256, 256, "uint4", 32, True, True, "quantized", None),
    (256, 256, 256, "uint4", 32, True, True, "quantized", False),
]


@pytest.mark.parametrize("case_args", REF_GEMM_CASES)
def test_reference_gemm_cases(case_args):
    M, N, K, wd, g, ws, wz, zm, fd = case_args
    case = make_case(M, N, K, W_dtype=wd, group_size=g, with_scaling=ws, with_zeros=wz, zeros_mode=zm,
                     fast_decoding=fd)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("M", [8, 16, 17, 33, 100, 128, 200])
@pytest.mark.parametrize("with_bias", [False, True])
def test_ragged_m_and_bias(M, with_bias):
    case = make_case(M, 384, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                     zeros_mode="original", with_bias=with_bias, seed=M)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("wd", ["uint2", "int2", "uint1", "int1", "uint8", "int8", "nf4", "fp4_e2m1"])
@pytest.mark.parametrize("fd", [None, False])
def test_other_weight_formats_fp16(wd, fd):
    if wd in ("nf4", "fp4_e2m1") and fd is False:
        pytest.skip("fast_decoding only exists for integer formats")
    case = make_case(64, 256, 512, W_dtype=wd, group_size=128, with_scaling=True, fast_decoding=fd, scale_mul=0.05)
    got, mm = hip_output(case)
    assert mm.plans[64]["kernel_family"] == 2
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("strict", [True, False])
def test_e4m3_weight_fp16_activation_gemm(strict):
    case = make_case(32, 256, 256, W_dtype="e4m3_float8", group_size=32, with_scaling=True)
    got, _ = hip_output(case, strict_reference=strict)
    assert_fp_parity(got, oracle_output(case, strict_reference=strict))


def test_dense_fp16_gemm():
    rng = np.random.default_rng(0)
    import bitblas_amd as bitblas
    A = (rng.random((96, 512), dtype=np.float32) - 0.5).astype(np.float16)
    W = (rng.random((256, 512), dtype=np.float32) - 0.5).astype(np.float16)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=96, N=256, K=512, A_dtype="float16", W_dtype="float16"),
                        enable_tuning=False)
    assert mm.plans[96]["kernel_family"] == 2
    out = mm(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()).cpu().numpy()
    want = (A.astype(np.float64) @ W.astype(np.float64).T).astype(np.float16)
    assert_fp_parity(out, want)


@pytest.mark.parametrize("M", [16, 256])
@pytest.mark.parametrize("wd,fd", [("int2", None), ("int2", False), ("int4", None), ("uint4", None), ("int1", None), ("int1", False)])
@pytest.mark.parametrize("out_dtype", ["int32", "float32"])
def test_int8_activation_gemm_exact(M, wd, fd, out_dtype):
    """BASELINE c4 family (BitNet W_int2 A_int8): int32 accumulation in the matrix core, bit exact."""
    case = make_case(M, 256, 1024, W_dtype=wd, A_dtype="int8", out_dtype=out_dtype, fast_decoding=fd)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    assert np.array_equal(got, oracle_output(case))


def test_int8_dense_gemm_exact():
    rng = np.random.default_rng(1)
    import bitblas_amd as bitblas
    A8 = rng.integers(-128, 128, size=(48, 512), dtype=np.int8)
    W8 = rng.integers(-128, 128, size=(256, 512), dtype=np.int8)
    mm8 = bitblas.Matmul(bitblas.MatmulConfig(M=48, N=256, K=512, A_dtype="int8", W_dtype="int8",
                                              accum_dtype="int32", out_dtype="int32"), enable_tuning=False)
    out8 = mm8(torch.from_numpy(A8).cuda(), torch.from_numpy(W8).cuda()).cpu().numpy()
    assert np.array_equal(out8, A8.astype(np.int64) @ W8.astype(np.int64).T)


def _whole_output_check(case, got, exact=False):
    """every element of the M x N output against the oracle (its product runs through a threaded float64 GEMM - exact for the
    integer paths, oracle/wqaa_oracle.py: exact_int_matmul - so a 4096^3 case costs seconds, and nothing is sampled)"""
    want = oracle_output(case)
    assert got.shape == want.shape
    if exact:
        assert np.array_equal(got, want)
    else:
        assert_fp_parity(got, want)


@pytest.mark.parametrize("M", [16, 128, 4096])
def test_baseline_c3_uint4_zeros_full_size_every_output_element(M):
    """BASELINE c3 at full size, all M x N elements against the oracle."""
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                     zeros_mode="original", scale_mul=0.02, seed=3)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    _whole_output_check(case, got)
    if M == 4096:
        # size-independent property: identical activation rows give identical output rows
        case2 = dict(case)
        A2 = case["A"].copy()
        A2[1::2] = A2[0::2]
        case2["A"] = A2
        got2, _ = hip_output(case2, matmul=mm)
        assert np.array_equal(got2[1::2], got2[0::2])
        assert np.array_equal(got2[0::2], got[0::2])


@pytest.mark.parametrize("M,N,K", [(1536, 4096, 1024), (2048, 4096, 512), (3000, 4096, 512), (1024, 11008, 512), (600, 11008, 512),
                                   (300, 22016, 256), (2500, 4104, 768), (1000, 2048, 1024), (129, 8192, 512), (777, 3000, 256)])
def test_prefill_sized_m_whichever_tile_the_selector_takes_every_output_element(M, N, K):
    """M between the decode batches and the full chip: the selector chooses between the 256-row and the 128-row ping-pong tile
    and the lockstep members by an estimate of the rounds each needs (csrc/wqaa_gemm.hip) - all of them against the oracle on
    the whole output."""
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.02, seed=M + N)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    _whole_output_check(case, got)


@pytest.mark.parametrize("M,N,K", [(2048, 4096, 1024), (1024, 4096, 512), (1100, 2048, 1536)])
def test_prefill_sized_m_int2_int8_bit_exact_every_output_element(M, N, K):
    case = make_case(M, N, K, W_dtype="int2", A_dtype="int8", out_dtype="int32", seed=M + K)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    _whole_output_check(case, got, exact=True)


@pytest.mark.parametrize("M,N,K,kw", [(128, 1024, 2048, dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.05)),
                                      (48, 768, 4096, dict(W_dtype="int2", A_dtype="int8", out_dtype="int32")),
                                      (1024, 1024, 1024, dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.05))])
def test_store_policy_of_partial_sums_and_large_outputs_does_not_change_a_bit(M, N, K, kw, monkeypatch):
    """Split-K partial sums and large output tiles leave the chip write-through (sc0 sc1 stores; csrc/wqaa_gemm.hip,
    `WQAA_GEMM_TUNE=ws_policy`): a cache policy, not arithmetic - plain stores must give the same bits."""
    case = make_case(M, N, K, seed=M + K, **kw)
    got, mm = hip_output(case)
    set_knobs(monkeypatch, "gemm", ws_policy="0")
    plain, mm2 = hip_output(case)
    assert mm.plans[M]["name"] == mm2.plans[M]["name"]
    assert np.array_equal(got, plain)
    set_knobs(monkeypatch, "gemm", ws_policy="19")
    forced, _ = hip_output(case)
    assert np.array_equal(got, forced)


def test_baseline_c4_int2_int8_gemm_full_size_every_output_element_bit_exact():
    case = make_case(4096, 4096, 4096, W_dtype="int2", A_dtype="int8", out_dtype="int32", seed=4)
    got, mm = hip_output(case)
    assert mm.plans[4096]["kernel_family"] == 2
    _whole_output_check(case, got, exact=True)


def test_gemv_and_gemm_agree_on_the_same_rows():
    """Dynamic-M operator: rows pushed through the GEMV family (m < 8) and the GEMM family must match
    to fp16 rounding of the fp32 accumulators."""
    import bitblas_amd as bitblas
    case = make_case(64, 512, 1024, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, seed=9)
    cfg = bitblas.MatmulConfig(M=[1, 64], N=512, K=1024, A_dtype="float16", W_dtype="uint4", group_size=128,
                               with_scaling=True, with_zeros=True)
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    W = mm.transform_weight(torch.from_numpy(case["w_user"]).cuda())
    sc, zr = torch.from_numpy(case["scale"]).cuda(), torch.from_numpy(case["zeros"]).cuda()
    A = torch.from_numpy(case["A"]).cuda()
    full = mm(A, W, scale=sc, zeros=zr).cpu().numpy()
    one = mm(A[:1], W, scale=sc, zeros=zr).cpu().numpy()
    assert_fp_parity(full, oracle_output(case))
    assert_fp_parity(one, full[:1], rtol=2e-3)


@pytest.mark.parametrize("M", [8, 64, 300])
@pytest.mark.parametrize("a_dt,w_dt", [("e4m3_float8", "e4m3_float8"), ("e4m3_float8", "e5m2_float8"),
                                       ("e5m2_float8", "e4m3_float8"), ("e5m2_float8", "e5m2_float8")])
def test_dense_fp8_gemm(M, a_dt, w_dt):
    """BASELINE c5 family: fp8 x fp8 -> fp32 on the fp8 matrix core (reference: test_general_matmul_fp8.py:11-59,
    which prints and never asserts - parity is against the exact OCP decode + fp64 matmul)."""
    import bitblas_amd as bitblas
    rng = np.random.default_rng(M)
    tdt = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}
    N, K = 256, 512
    A8 = torch.from_numpy((rng.random((M, K), dtype=np.float32) * 2 - 1)).to(tdt[a_dt])
    W8 = torch.from_numpy((rng.random((N, K), dtype=np.float32) * 2 - 1)).to(tdt[w_dt])
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=a_dt, W_dtype=w_dt, accum_dtype="float32",
                                             out_dtype="float32"), enable_tuning=False)
    assert mm.plans[M]["kernel_family"] == 2
    out = mm(A8.cuda(), W8.cuda()).cpu().numpy()
    want = oracle.matmul_dense(A8.view(torch.int8).numpy(), W8.view(torch.int8).numpy(), a_dtype=a_dt, w_dtype=w_dt,
                               out_dtype="float32")
    assert_fp_parity(out, want, rtol=1e-4, atol_frac=1e-4)   # fp32 accumulation order inside the matrix core


def test_dense_fp8_gemv_matches_gemm():
    import bitblas_amd as bitblas
    rng = np.random.default_rng(3)
    N, K = 256, 1024
    A8 = torch.from_numpy((rng.random((8, K), dtype=np.float32) * 2 - 1)).to(torch.float8_e4m3fn)
    W8 = torch.from_numpy((rng.random((N, K), dtype=np.float32) * 2 - 1)).to(torch.float8_e4m3fn)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=[1, 8], N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8",
                                             accum_dtype="float32", out_dtype="float32"), enable_tuning=False)
    assert mm.plans[1]["kernel_family"] == 1 and mm.plans[8]["kernel_family"] == 2
    full = mm(A8.cuda(), W8.cuda()).cpu().numpy()
    one = mm(A8[:1].cuda(), W8.cuda()).cpu().numpy()
    want = oracle.matmul_dense(A8.view(torch.int8).numpy(), W8.view(torch.int8).numpy(), a_dtype="e4m3_float8",
                               out_dtype="float32")
    assert_fp_parity(full, want, rtol=1e-4, atol_frac=1e-4)
    assert_fp_parity(one, want[:1], rtol=1e-4, atol_frac=1e-4)


def _bf16_case(M, N, K, W_dtype, g, with_scaling, zeros_mode=None, seed=0):
    """A_dtype = bfloat16 case builder (reference: testing/python/operators/test_general_matmul_bf16.py:56-178)."""
    import bitblas_amd as bitblas
    rng = np.random.default_rng(seed)
    src, bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[W_dtype]
    A = (torch.from_numpy(rng.random((M, K), dtype=np.float32)) - 0.5).to(torch.bfloat16)
    if src == "uint":
        w_user = rng.integers(0, 1 << bit, size=(N, K)).astype(np.int8) if bit < 8 else rng.integers(0, 128, size=(N, K)).astype(np.int8)
        codes = w_user
    elif src in ("nf", "fp"):
        codes = w_user = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    elif src == "fp_e4m3":
        w8 = torch.from_numpy((rng.random((N, K), dtype=np.float32) * 2 - 1)).to(torch.float8_e4m3fn)
        codes = w_user = w8.view(torch.int8).numpy()
    else:
        maxq = 1 << (bit - 1)
        w_user = rng.integers(-maxq, maxq, size=(N, K)).astype(np.int8)
        codes = (w_user + maxq).astype(np.int8) if bit < 8 else w_user
    gg = K if g == -1 else g
    scale = zeros = None
    if with_scaling:
        scale = torch.from_numpy(rng.standard_normal((N, K // gg)).astype(np.float32) * 0.05).to(torch.bfloat16)
    if zeros_mode == "quantized":
        zint = np.clip((1 << (bit - 1)) + rng.integers(-2, 2, size=(K // gg, N)), 0, (1 << bit) - 1).astype(np.int8)
        zeros = oracle.general_compress(zint, bit)
    elif zeros_mode in ("original", "rescale"):
        # bfloat16 zero points: integers as GPTQ gives them, some with a fraction (then `w - z` itself rounds to bfloat16)
        z = ((1 << (bit - 1)) + rng.integers(-2, 3, size=(N, K // gg))).astype(np.float32)
        z[::3, 0] += 0.3125
        zt = torch.from_numpy(z).to(torch.bfloat16)
        if zeros_mode == "rescale":
            zt = (zt.float() * scale.float()).to(torch.bfloat16)
        zeros = zt
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="bfloat16", W_dtype=W_dtype, accum_dtype="float32", out_dtype="float32",
                               group_size=g, with_scaling=with_scaling, with_zeros=zeros is not None,
                               zeros_mode=zeros_mode or "original")
    mm = bitblas.Matmul(cfg, enable_tuning=False)
    assert cfg.fast_decoding is False                 # the reference's legalisation: no LOP3 layout for bf16
    Wt = mm.weight_transform(torch.from_numpy(codes)).cuda() if mm.weight_transform is not None else torch.from_numpy(codes).cuda()
    zdev = None if zeros is None else (zeros.cuda() if isinstance(zeros, torch.Tensor) else torch.from_numpy(zeros).cuda())
    out = mm(A.cuda(), Wt, scale=None if scale is None else scale.cuda(), zeros=zdev).cpu().numpy()
    want = oracle.matmul_dequant(A.float().numpy(), codes, source_format=src, bit=bit,
                                 scale=None if scale is None else scale.float().numpy(),
                                 zeros=zeros.float().numpy() if isinstance(zeros, torch.Tensor) else zeros,
                                 zeros_mode=zeros_mode or "original", group_size=gg, a_dtype="bfloat16", out_dtype="float32")
    return out, want, mm


@pytest.mark.parametrize("M", [1, 3, 64, 1024])
@pytest.mark.parametrize("W_dtype,g,ws,zm", [("uint4", -1, False, None), ("uint4", 32, True, None), ("int4", 128, True, None),
                                             ("uint4", 128, True, "quantized"), ("uint2", 128, True, None), ("int8", 128, True, None),
                                             ("uint4", 128, True, "original"), ("uint4", 128, True, "rescale"),
                                             ("uint2", 64, True, "original"), ("uint8", 128, True, "rescale"), ("uint1", 128, True, "original")])
def test_bf16_activations(M, W_dtype, g, ws, zm):
    """reference cases test_general_matmul_bf16.py:170-178 (M in {1, 1024}, uint4, +-scale g=32) and neighbours."""
    if W_dtype == "uint4" and g == 32 and M >= 8:
        g = 32   # GEMM lane chunk is 32 elements: g = 32 is legal
    out, want, mm = _bf16_case(M, 512, 1024, W_dtype, g, ws, zm, seed=M)
    assert mm.plans[M]["kernel_family"] == (1 if M < 3 else 2)
    assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)


def test_bf16_dense():
    import bitblas_amd as bitblas
    rng = np.random.default_rng(1)
    for M in (2, 96):
        A = (torch.from_numpy(rng.random((M, 512), dtype=np.float32)) - 0.5).to(torch.bfloat16)
        W = (torch.from_numpy(rng.random((256, 512), dtype=np.float32)) - 0.5).to(torch.bfloat16)
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=256, K=512, A_dtype="bfloat16", W_dtype="bfloat16", accum_dtype="float32",
                                                 out_dtype="float32"), enable_tuning=False)
        out = mm(A.cuda(), W.cuda()).cpu().numpy()
        want = (A.double().numpy() @ W.double().numpy().T).astype(np.float32)
        assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)


@pytest.mark.parametrize("M", [1, 5])
def test_small_groups_fall_through_to_the_mfma_family(M):
    """uint2 with group_size 32: below the GEMV's 64-element lane chunk, inside the GEMM's 32-element one."""
    case = make_case(M, 256, 512, W_dtype="uint2", group_size=32, with_scaling=True, with_zeros=True, zeros_mode="original",
                     scale_mul=0.05, seed=M)
    got, mm = hip_output(case)
    assert mm.plans[M]["kernel_family"] == 2
    assert_fp_parity(got, oracle_output(case))


@pytest.mark.parametrize("M,N", [(5, 1024), (8, 1024), (9, 3584), (13, 4096), (16, 4096)])
@pytest.mark.parametrize("kw", [dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.05),
                                dict(W_dtype="int2", A_dtype="int8", out_dtype="int32")])
def test_decode_batch_member_against_the_split_k_member(M, N, kw, monkeypatch):
    """M = 3..8 (and 9..16 when the grid is about one weight fragment per CU) runs the one-launch member (K split
    across the waves of a workgroup, summed in LDS); the split-K skinny member + reduce launch must agree: bit
    exact for integers, within the fp16 bound for floats (different summation order)."""
    case = make_case(M, N, 4096 if N <= 1024 else 2048, seed=M, **kw)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdl")          # activations through LDS-DMA
    set_knobs(monkeypatch, "gemm", decode="0")
    got2, mm2 = hip_output(case)
    assert mm2.plans[M]["name"].endswith("xs")
    want = oracle_output(case)
    if kw.get("A_dtype") == "int8":
        assert np.array_equal(got, got2) and np.array_equal(got, want)
    else:
        assert_fp_parity(got, want)
        assert_fp_parity(got2, want)


@pytest.mark.parametrize("M", [65, 100, 128])
@pytest.mark.parametrize("kw", [dict(W_dtype="uint4", with_zeros=True, zeros_mode="original"),
                                dict(W_dtype="uint4", with_zeros=True, zeros_mode="rescale"),
                                dict(W_dtype="int4", with_zeros=True, zeros_mode="original")])
@pytest.mark.parametrize("N,K", [(512, 1024), (200, 2048)])
def test_block_metadata_member_is_bit_identical(M, kw, N, K, monkeypatch):
    """g = 128 with K / g a multiple of 4: the 64-row member fetches Scale / Zeros of four k-steps with one
    8-byte load each (plan suffix xw).  Same arithmetic in the same order as the per-step form: bit identical,
    also under split-K (a slice may start inside a block of four)."""
    case = make_case(M, N, K, seed=M + N, group_size=128, with_scaling=True, scale_mul=0.05, **kw)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xw")
    set_knobs(monkeypatch, "gemm", wide="0")
    got2, mm2 = hip_output(case)
    assert not mm2.plans[M]["name"].endswith("xw")
    assert np.array_equal(got.view(np.uint16), got2.view(np.uint16))
    assert_fp_parity(got, oracle_output(case))
    set_knobs(monkeypatch, "gemm", wide=None)
    for ks in (3, 5):
        set_knobs(monkeypatch, "gemm", ksplit=str(ks))
        got3, mm3 = hip_output(case)
        assert mm3.plans[M]["name"].endswith("xw")
        assert_fp_parity(got3, oracle_output(case))


@pytest.mark.parametrize("M", [1, 4, 64, 300])
@pytest.mark.parametrize("W_dtype,g,ws", [("nf4", -1, False), ("nf4", 128, True), ("fp4_e2m1", -1, False), ("fp4_e2m1", 64, True),
                                          ("e4m3_float8", -1, False), ("e4m3_float8", 128, True), ("uint1", 128, True)])
def test_bf16_activations_other_weight_formats(M, W_dtype, g, ws):
    """the BF16 rows of the reference's support matrix (README.md:63-70) beyond the integer formats: NF4 (table
    in bfloat16, general_matmul/__init__.py:413-434), FP4_E2M1, FP8_E4M3 (exact decode) and UINT1 on the MFMA path"""
    out, want, mm = _bf16_case(M, 512, 1024, W_dtype, g, ws, None, seed=M)
    assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)


def test_split_k_operator_matches_matmul():
    """`MatmulWithSplitK` (ref: bitblas/ops/general_matmul_splitk.py:26-199): same result as
    `Matmul` on the same operands - split-K is the selector's decision here, k_split a hint"""
    import bitblas_amd as bitblas
    case = make_case(16, 1024, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.05)
    got, mm = hip_output(case)
    c = case["config"]
    cfg = bitblas.MatmulConfigWithSplitK(M=16, N=1024, K=4096, A_dtype="float16", W_dtype="uint4", group_size=128,
                                         with_scaling=True, with_zeros=True, zeros_mode=c.zeros_mode, k_split=4)
    got_sk, mm_sk = hip_output(case, matmul=bitblas.MatmulWithSplitK(cfg, enable_tuning=False))
    assert np.array_equal(got, got_sk)
    assert_fp_parity(got_sk, oracle_output(case))


@pytest.mark.parametrize("M,ks", [(128, 2), (128, 5), (200, 3), (1, 2), (2, 4)])
def test_k_split_request_is_honoured_and_exact_enough(M, ks):
    """the caller's k_split sets the split-K count of the pipelined members / the in-workgroup K split of the M <= 2
    exact-product GEMV; the result meets the same parity bound whatever the split"""
    import bitblas_amd as bitblas
    K = 4096 if M > 2 else 16384          # the 4-bit GEMV steps are 4096 deep
    case = make_case(M, 1024, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.05, seed=ks)
    c = case["config"]
    cfg = bitblas.MatmulConfigWithSplitK(M=M, N=1024, K=K, A_dtype="float16", W_dtype="uint4", group_size=128,
                                         with_scaling=True, with_zeros=True, zeros_mode=c.zeros_mode, k_split=ks)
    mm = bitblas.MatmulWithSplitK(cfg, enable_tuning=False, strict_reference=M > 2)
    got, _ = hip_output(case, matmul=mm)
    assert mm.plans[M]["split_k"] == ks, mm.plans[M]
    assert_fp_parity(got, oracle_output(case), **case_contract(case, default_members=M <= 2, m=M))
