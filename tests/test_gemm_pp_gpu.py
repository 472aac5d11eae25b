"""The ping-pong MFMA members (csrc/wqaa_gemm_pp_kernel.h: 256 x 256 and 128 x 256 tiles) against the CPU oracle.

The selector takes them for large M by an estimate of the rounds of the chip each tile needs; `WQAA_GEMM_TUNE=pp_tile=256 / 128` (a
plan-time tuning aid) pins the tile so that its edge cases run at sizes the oracle finishes in seconds: ragged M and N, one trip
of the main loop, every dequant mode it implements, integer and non-integer zero points (two decode paths), both checkpoint
layouts, bias.  BASELINE c3 / c4 at full size run through the 256-row tile in tests/test_gemm_gpu.py."""
import numpy as np
import pytest
import torch

from helpers import knob_value, set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[256, 128], ids=["tile256", "tile128"])
def pin_the_tile(monkeypatch, request):
    set_knobs(monkeypatch, "gemm", pp_tile=str(request.param))
    return request.param


def _run(case, M, exact=False):
    import os
    got, mm = hip_output(case)
    plan = mm.plans[M]
    assert plan["kernel_family"] == 2 and plan["name"].endswith("pp"), plan["name"]
    assert f"_tcx{knob_value('gemm', 'pp_tile')}x256x" in plan["name"], plan["name"]
    want = oracle_output(case)
    if exact:
        assert np.array_equal(got, want)
    else:
        assert_fp_parity(got, want)
    return got, mm


@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (256, 256, 256), (513, 264, 1024)])
@pytest.mark.parametrize("zeros_mode", ["original", "rescale"])
def test_uint4_scale_zeros(M, N, K, zeros_mode):
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode, seed=M + K)
    _run(case, M)


@pytest.mark.parametrize("M,N,K", [(300, 544, 512), (256, 256, 256), (513, 288, 1024)])
@pytest.mark.parametrize("wd", ["uint4", "int4"])
def test_4bit_packed_integer_zero_points(M, N, K, wd):
    """zeros_mode = "quantized" (GPTQ checkpoints: Zeros packed 4-bit along N, one row per group), N in whole waves of 32 rows"""
    case = make_case(M, N, K, W_dtype=wd, group_size=128, with_scaling=True, with_zeros=True, zeros_mode="quantized", seed=M + N)
    _run(case, M)


@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (257, 264, 1024)])
@pytest.mark.parametrize("zeros_mode", [None, "original", "rescale", "quantized"])
def test_uint4_one_group_per_row(M, N, K, zeros_mode):
    """group_size = -1 (the reference's default): per-channel Scale / Zeros - a row's 16-byte window opens on the even element in
    front of an odd row (ragged N: the last rows' windows are clamped into the array)"""
    if zeros_mode == "quantized":
        N = (N + 31) // 32 * 32
    case = make_case(M, N, K, W_dtype="uint4", group_size=-1, with_scaling=True, with_zeros=zeros_mode is not None,
                     zeros_mode=zeros_mode or "original", seed=M + N + K)
    _run(case, M)


@pytest.mark.parametrize("M,N,K", [(300, 544, 512), (513, 288, 1024)])
@pytest.mark.parametrize("wd,g,ws,zm", [("uint4", -1, False, None), ("int4", 128, True, None), ("uint4", 128, True, "original"),
                                        ("uint4", 128, True, "rescale"), ("uint4", 128, True, "quantized"), ("uint4", -1, True, "original"),
                                        ("nf4", 128, True, None), ("fp4_e2m1", -1, False, None)])
def test_bfloat16_activations(M, N, K, wd, g, ws, zm):
    """A_dtype = bfloat16 (float32 accumulate, float32 output: the reference's test_general_matmul_bf16.py configuration): the
    lockstep member's per-word decode inside the ping-pong loop, bfloat16 MFMA"""
    from test_gemm_gpu import _bf16_case
    out, want, mm = _bf16_case(M, N, K, wd, g, ws, zm, seed=M + N)
    import os
    assert mm.plans[M]["name"].endswith("pp") and f"_tcx{knob_value('gemm', 'pp_tile')}x256x" in mm.plans[M]["name"], mm.plans[M]["name"]
    assert_fp_parity(out, want, rtol=1e-5, atol_frac=1e-5)


def test_float32_output_of_the_float16_members():
    case = make_case(300, 520, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, out_dtype="float32", seed=9)
    _run(case, 300)


def test_uint4_fractional_zero_points_take_the_general_decode():
    case = make_case(300, 520, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, seed=5)
    case["zeros"] = (case["zeros"].astype(np.float32) + 0.375).astype(np.float16)
    _run(case, 300)
    # one fractional zero point in one row of one wave is enough to leave the integer path for that wave only
    case2 = make_case(300, 520, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, seed=6)
    case2["zeros"][37, 2] = np.float16(7.25)
    _run(case2, 300)


@pytest.mark.parametrize("wd", ["int4", "uint4"])
@pytest.mark.parametrize("g", [128, 256, 512])
def test_int4_scale_only_group_sizes(wd, g):
    case = make_case(260, 512, 1024, W_dtype=wd, group_size=g, with_scaling=True, seed=g)
    _run(case, 260)


@pytest.mark.parametrize("wd", ["int4", "uint4", "nf4", "fp4_e2m1"])
def test_no_scaling(wd):
    case = make_case(257, 256, 512, W_dtype=wd, seed=11)
    _run(case, 257)


@pytest.mark.parametrize("wd", ["nf4", "fp4_e2m1"])
def test_lut_formats_with_scaling(wd):
    case = make_case(300, 512, 512, W_dtype=wd, group_size=128, with_scaling=True, seed=12)
    _run(case, 300)


def test_plain_layout_and_bias():
    case = make_case(300, 520, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, with_bias=True,
                     fast_decoding=False, seed=13)
    _run(case, 300)
    case = make_case(300, 520, 512, W_dtype="int4", group_size=128, with_scaling=True, with_bias=True, seed=14)
    _run(case, 300)


@pytest.mark.parametrize("wd", ["int2", "uint2"])
@pytest.mark.parametrize("fd", [None, False])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (256, 256, 1024)])
def test_int2_int8_bit_exact(wd, fd, M, N, K):
    case = make_case(M, N, K, W_dtype=wd, A_dtype="int8", out_dtype="int32", fast_decoding=fd, seed=M)
    _run(case, M, exact=True)


def test_int2_int8_bias_bit_exact():
    case = make_case(300, 520, 512, W_dtype="int2", A_dtype="int8", out_dtype="int32", with_bias=True, seed=15)
    _run(case, 300, exact=True)


@pytest.mark.parametrize("a_dt,w_dt", [("e4m3_float8", "e4m3_float8"), ("e5m2_float8", "e4m3_float8"), ("e4m3_float8", "e5m2_float8"),
                                       ("e5m2_float8", "e5m2_float8")])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (256, 256, 128), (513, 264, 1152)])
def test_dense_fp8_pairings_ragged(a_dt, w_dt, M, N, K, pin_the_tile):
    """BASELINE c5's kernel (both tiles) at sizes where the oracle checks EVERY element: ragged M / N, one k-tile, all four fp8
    pairings"""
    import bitblas_amd as bitblas
    import wqaa_oracle as oracle
    tdt = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    A = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).to(tdt[a_dt])
    W = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).to(tdt[w_dt])
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=a_dt, W_dtype=w_dt, accum_dtype="float32", out_dtype="float16"),
                        enable_tuning=False)
    assert mm.plans[M]["name"].endswith("pp") and f"_tcx{pin_the_tile}x256x" in mm.plans[M]["name"], mm.plans[M]["name"]
    out = mm(A, W)
    torch.cuda.synchronize()
    want = oracle.matmul_dense(A.view(torch.int8).cpu().numpy(), W.view(torch.int8).cpu().numpy(), a_dtype=a_dt, w_dtype=w_dt,
                               out_dtype="float32")
    assert_fp_parity(out.float().cpu().numpy(), want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-4)


@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (256, 256, 128), (513, 264, 1152)])
def test_dense_int8_bit_exact(M, N, K, with_bias, pin_the_tile):
    """INT8 x INT8 -> INT32 (README.md support matrix; tilelang/dense/matmul_mma.py) on the dense skeleton (round 4): bit exact over
    the full int8 range, ragged M / N, one k-tile, both tiles, int8 bias"""
    import bitblas_amd as bitblas
    import wqaa_oracle as oracle
    rng = np.random.default_rng(M + K)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    W = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    b = rng.integers(-8, 8, size=(N,), dtype=np.int8) if with_bias else None
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int8", W_dtype="int8", accum_dtype="int32", out_dtype="int32", with_bias=with_bias),
                        enable_tuning=False)
    assert mm.plans[M]["name"].endswith("pp") and f"_tcx{pin_the_tile}x256x128" in mm.plans[M]["name"], mm.plans[M]["name"]
    out = mm(torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda(), bias=None if b is None else torch.from_numpy(b).cuda())
    torch.cuda.synchronize()
    want = A.astype(np.int64) @ W.astype(np.int64).T
    if b is not None:
        want = want + b.astype(np.int64)[None, :]
    assert np.array_equal(out.cpu().numpy().astype(np.int64), want)


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (256, 256, 64), (513, 264, 1152)])
def test_dense_16bit_pairs_ragged(dt, M, N, K, pin_the_tile):
    """the reference's plain matmul (W_dtype == A_dtype, float16 / bfloat16: README.md support matrix, first rows;
    tilelang/dense/matmul_mma.py) on the dense skeleton with 16-bit lines (round 4) - also the second pass of the two-pass member:
    every element against the oracle, ragged M / N, one k-tile, both tiles"""
    import bitblas_amd as bitblas
    import wqaa_oracle as oracle
    tdt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[dt]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    A = (torch.rand((M, K), device="cuda", generator=gen) - 0.5).to(tdt)
    W = (torch.rand((N, K), device="cuda", generator=gen) - 0.5).to(tdt)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=dt, W_dtype=dt, accum_dtype="float32", out_dtype=dt), enable_tuning=False)
    assert mm.plans[M]["name"].endswith("pp") and f"_tcx{pin_the_tile}x256x64" in mm.plans[M]["name"], mm.plans[M]["name"]
    out = mm(A, W)
    torch.cuda.synchronize()
    want = oracle.matmul_dense(A.float().cpu().numpy(), W.float().cpu().numpy(), a_dtype=dt, w_dtype=dt, out_dtype="float32")
    if dt == "bfloat16":
        assert_fp_parity(out.float().cpu().numpy(), want, rtol=8e-3, atol_frac=8e-3)       # the bfloat16 result itself is rounded to 2^-8 relative
    else:
        assert_fp_parity(out.float().cpu().numpy(), want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-4)


@pytest.mark.parametrize("a_dt,w_dt", [("e4m3_float8", "e4m3_float8"), ("e5m2_float8", "e4m3_float8"), ("e4m3_float8", "e5m2_float8"),
                                       ("e5m2_float8", "e5m2_float8")])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (129, 128, 128), (513, 264, 1152), (1000, 136, 2048)])
def test_dense_fp8_128x128_tile(a_dt, w_dt, M, N, K, monkeypatch, pin_the_tile):
    """the dense fp8 member for outputs too small to give every CU a wider tile (the N / 8 column shards of BASELINE c5): wave grid
    4 x 2, one phase per k-tile - every element against the oracle, ragged M / N, all four pairings"""
    if pin_the_tile != 128:
        pytest.skip("one tile: runs once")
    import bitblas_amd as bitblas
    import wqaa_oracle as oracle
    set_knobs(monkeypatch, "gemm", pp_tile="128")
    set_knobs(monkeypatch, "gemm", pp_bn="128")
    tdt = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    A = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).to(tdt[a_dt])
    W = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).to(tdt[w_dt])
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=a_dt, W_dtype=w_dt, accum_dtype="float32", out_dtype="float16"),
                        enable_tuning=False)
    assert mm.plans[M]["name"].endswith("_tcx128x128x128pp"), mm.plans[M]["name"]
    out = mm(A, W)
    torch.cuda.synchronize()
    want = oracle.matmul_dense(A.view(torch.int8).cpu().numpy(), W.view(torch.int8).cpu().numpy(), a_dtype=a_dt, w_dtype=w_dt,
                               out_dtype="float32")
    assert_fp_parity(out.float().cpu().numpy(), want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-4)


def test_what_the_member_does_not_cover_falls_back(monkeypatch):
    """bfloat16 activations with 2-bit weights, bfloat16 output of float16 activations... - formats without a ping-pong member; an odd
    number (> 1) of groups per row, quantized zeros with N off the 32-row grid, K off the 256 grid: the lockstep member."""
    import bitblas_amd as bitblas
    for kw in (dict(N=512, A_dtype="bfloat16", out_dtype="bfloat16", accum_dtype="float32", W_dtype="uint2", group_size=128, with_scaling=True),
               dict(N=512, A_dtype="bfloat16", out_dtype="float32", accum_dtype="float32", W_dtype="int8", group_size=128, with_scaling=True),
               dict(N=520, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="quantized"),
               dict(N=512, K=768, A_dtype="float16", W_dtype="uint4", group_size=256, with_scaling=True)):      # three groups per row
        kw.setdefault("K", 512)
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=512, **kw), enable_tuning=False)
        assert not mm.plans[512]["name"].endswith("pp"), mm.plans[512]["name"]
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=512, N=512, K=384, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True),
                        enable_tuning=False)
    assert not mm.plans[512]["name"].endswith("pp")


def test_two_launches_of_different_rows_share_nothing():
    """identical activation rows give identical output rows (each row of C depends on its own row of A only), also across the
    two wave groups and the 16-row fragments of a tile"""
    case = make_case(512, 512, 512, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, seed=21)
    A = case["A"].copy()
    A[1::2] = A[0::2]
    case["A"] = A
    got, _ = _run(case, 512)
    assert np.array_equal(got[1::2], got[0::2])


@pytest.mark.parametrize("kind", ["float16", "bfloat16", "int8", "e4m3_float8", "e5m2_e4m3", "e4m3_e5m2", "e5m2_float8"])
@pytest.mark.parametrize("M,N,K", [(300, 520, 512), (512, 512, 1152), (257, 264, 128)])
def test_dense_2x4_wave_grid_is_bit_identical_to_the_1x8_grid(kind, M, N, K, monkeypatch, pin_the_tile):
    """round 5: the dense 256 x 256 tile on a 2 (m) x 4 (n) wave grid (wq_gemm_pp8w_kernel: both operands shared LDS tiles, a third
    fewer LDS reads per MFMA) adds every output's products in the same order as the 1 x 8 grid it replaces (WQAA_GEMM_TUNE=pp8_wide=0)"""
    if pin_the_tile != 256:
        pytest.skip("the 256-row tile only")
    import bitblas_amd as bitblas
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M + K)
    if kind == "int8":
        A = torch.randint(-128, 128, (M, K), device="cuda", dtype=torch.int8, generator=gen)
        W = torch.randint(-128, 128, (N, K), device="cuda", dtype=torch.int8, generator=gen)
        cfg = dict(A_dtype="int8", W_dtype="int8", accum_dtype="int32", out_dtype="int32")
    elif kind in ("float16", "bfloat16"):
        tdt = torch.float16 if kind == "float16" else torch.bfloat16
        A = (torch.rand((M, K), device="cuda", generator=gen) - 0.5).to(tdt)
        W = (torch.rand((N, K), device="cuda", generator=gen) - 0.5).to(tdt)
        cfg = dict(A_dtype=kind, W_dtype=kind, accum_dtype="float32", out_dtype=kind)
    else:
        a_dt, w_dt = {"e4m3_float8": ("e4m3_float8", "e4m3_float8"), "e5m2_e4m3": ("e5m2_float8", "e4m3_float8"),
                      "e4m3_e5m2": ("e4m3_float8", "e5m2_float8"), "e5m2_float8": ("e5m2_float8", "e5m2_float8")}[kind]
        tdt = {"e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}
        A = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).to(tdt[a_dt])
        W = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).to(tdt[w_dt])
        cfg = dict(A_dtype=a_dt, W_dtype=w_dt, accum_dtype="float32", out_dtype="float16")
    outs = []
    for wide in ("1", "0"):
        set_knobs(monkeypatch, "gemm", pp8_wide=wide)
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, **cfg), enable_tuning=False)
        assert mm.plans[M]["name"].endswith("pp"), mm.plans[M]["name"]
        out = mm(A, W)
        torch.cuda.synchronize()
        outs.append(out.view(torch.int16 if out.element_size() == 2 else torch.int32).cpu())
    assert torch.equal(outs[0], outs[1])
