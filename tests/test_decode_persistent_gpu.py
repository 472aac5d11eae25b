"""The persistent form of the one-launch decode member (csrc/wqaa_gemm_kernel.h wq_gemm_decode_lds_kernel, plan suffix `xdlp`;
round 4): decode batches M = 3 ... 16 on outputs wider than the chip has CUs x 16 rows (11008, 12288, 22016 x 4096: the MLP and
q/k/v shapes of a 7B model) used to take the split-K skinny member + its reduce launch (13-15 us); now a grid of one workgroup
per CU stages the activations once and walks the 16-row weight fragments with the next one's weights in flight.
Checked against the CPU oracle (the reference's decode batches: `Matmul` with M in its default opt list [1, 16, 32, ...],
ops/general_matmul/__init__.py:188-192; scheduler choice matmul_dequantize.py:93-109) and, bit for bit, against the
one-fragment-per-workgroup form of the same kernel (same k ranges per wave, same order of the final sum)."""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
from helpers import set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


def _both(case, M, monkeypatch, exact=False):
    set_knobs(monkeypatch, "gemm", decode_persist=None)
    set_knobs(monkeypatch, "gemm", decode_force=None)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlp"), mm.plans[M]["name"]
    assert mm.plans[M]["grid"] < (case["N"] + 15) // 16
    want = oracle_output(case)
    if exact:
        assert np.array_equal(got, want)
    else:
        assert_fp_parity(got, want)
    set_knobs(monkeypatch, "gemm", decode_persist="0")
    set_knobs(monkeypatch, "gemm", decode_force="1")
    one, mm1 = hip_output(case)
    assert mm1.plans[M]["name"].endswith("xdl"), mm1.plans[M]["name"]
    assert np.array_equal(got.view(np.uint16) if got.dtype == np.float16 else got, one.view(np.uint16) if one.dtype == np.float16 else one)


@pytest.mark.parametrize("M", [3, 5, 16])
@pytest.mark.parametrize("N,K", [(4112, 512), (11008, 4096), (6000, 2048)])
def test_uint4_scale_zeros(M, N, K, monkeypatch):
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.05, seed=M + N)
    _both(case, M, monkeypatch)


@pytest.mark.parametrize("M", [4, 16])
@pytest.mark.parametrize("N", [12928, 16544, 22016])
def test_four_to_six_fragments_per_workgroup(M, N, monkeypatch):
    """the hand-counted form's second batch (the first batch's registers refilled as they are consumed): 3.2 / 4.04 / 5.4 rounds of fragments"""
    case = make_case(M, N, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.05, seed=M + N)
    _both(case, M, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["rescale", "quantized"])
def test_other_zero_point_forms_and_bias(zeros_mode, monkeypatch):
    case = make_case(9, 8192 + 32, 1024, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode, with_bias=True,
                     scale_mul=0.05, seed=11)
    _both(case, 9, monkeypatch)


@pytest.mark.parametrize("wd", ["int4", "nf4", "uint2"])
def test_other_weight_formats(wd, monkeypatch):
    case = make_case(12, 4096 + 64, 1024, W_dtype=wd, group_size=128, with_scaling=True, scale_mul=0.05, seed=13)
    _both(case, 12, monkeypatch)


def test_formats_and_widths_that_measured_no_better_keep_their_members():
    """int2 x int8 and outputs beyond three rounds of fragments (22016 x 4096): the skinny member + reduce stays (profiles/r04_ab_decode_persistent.txt)"""
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=8, N=11008, K=4096, A_dtype="int8", W_dtype="int2", accum_dtype="int32", out_dtype="int32"), enable_tuning=False)
    assert not mm.plans[8]["name"].endswith("xdlp"), mm.plans[8]["name"]
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=8, N=22016, K=4096, A_dtype="float16", W_dtype="uint4", group_size=-1, with_scaling=True), enable_tuning=False)
    assert not mm.plans[8]["name"].endswith("xdlp"), mm.plans[8]["name"]       # (per-channel scales: the compiler-tracked form, three rounds at most)


@pytest.mark.parametrize("M", [4, 16])
def test_dense_fp8_decode_batches(M, monkeypatch):
    """e4m3 x e4m3 (BASELINE c5 shapes at decode batches): the same kernel family, K = 4096 fits one block per wave"""
    import wqaa_oracle as oracle
    N, K = 8192, 4096
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M)
    A = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
    W = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
    set_knobs(monkeypatch, "gemm", decode_persist=None)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16"),
                        enable_tuning=False)
    out = mm(A, W)
    torch.cuda.synchronize()
    want = oracle.matmul_dense(A.view(torch.int8).cpu().numpy(), W.view(torch.int8).cpu().numpy(), a_dtype="e4m3_float8", w_dtype="e4m3_float8", out_dtype="float32")
    assert_fp_parity(out.float().cpu().numpy(), want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-4)
    if mm.plans[M]["name"].endswith("xdlp"):
        set_knobs(monkeypatch, "gemm", decode_persist="0")
        set_knobs(monkeypatch, "gemm", decode_force="1")
        mm1 = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16"),
                             enable_tuning=False)
        assert torch.equal(out, mm1(A, W))


def test_long_rows_keep_the_one_fragment_form():
    """K beyond one block per wave (4096 x 11008): the activations of a wave do not stay staged - one fragment per workgroup as before"""
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=8, N=8192, K=11008, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True), enable_tuning=False)
    assert not mm.plans[8]["name"].endswith("xdlp"), mm.plans[8]["name"]


def _long_both(case, M, monkeypatch, want_long=True):
    """whole-tile form (`xdlt`) against the oracle and, bit for bit, against the block-by-block form of the same kernel"""
    set_knobs(monkeypatch, "gemm", decode_persist=None, decode_force=None, decode_long=None)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlt") == want_long, mm.plans[M]["name"]
    assert_fp_parity(got, oracle_output(case))
    set_knobs(monkeypatch, "gemm", decode_long="0")
    set_knobs(monkeypatch, "gemm", decode_force="1")
    one, mm1 = hip_output(case)
    assert mm1.plans[M]["name"].endswith("xdl"), mm1.plans[M]["name"]
    assert np.array_equal(got.view(np.uint16), one.view(np.uint16))


@pytest.mark.parametrize("M,N,K", [(3, 8192, 11008), (4, 4096 + 48, 11008), (4, 8192 - 16, 12288),                      # three blocks per wave: M <= 4, two fragments
                                   (5, 4096 + 16, 8192), (8, 8192, 8192), (7, 10240, 8192), (8, 3 * 4096, 8192), (3, 6144, 6144)])   # two: M <= 8, three fragments
def test_whole_tile_form_on_long_k(M, N, K, monkeypatch):
    """K > 4096: the wave's k-range is 2 ... 4 blocks, its activations fit in M-sized slots, units (fragment, block) are walked three
    in flight (K / g = 86 at 11008 is even, not a multiple of 4: the last block's metadata load ends inside the row)"""
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.05, seed=M + N + K)
    _long_both(case, M, monkeypatch)


@pytest.mark.parametrize("zeros_mode,wd", [("rescale", "uint4"), ("original", "int4"), (None, "nf4")])
def test_whole_tile_form_other_formats(zeros_mode, wd, monkeypatch):
    case = make_case(6, 8192 + 32, 8192, W_dtype=wd, group_size=128, with_scaling=True, with_zeros=zeros_mode is not None, zeros_mode=zeros_mode or "original",
                     with_bias=True, scale_mul=0.05, seed=29)
    _long_both(case, 6, monkeypatch)


def test_whole_tile_form_keeps_to_what_fits(monkeypatch):
    """M = 9 at K = 8192 (3 KiB slots x 8 k-steps), M = 5 at K = 11008 and seven blocks per wave (K = 28672) do not fit a 16 KiB region;
    one fragment per workgroup measured no better than block by block and keeps the old form; four blocks allow one fragment only"""
    for M, N, K in ((9, 8192, 8192), (5, 8192, 11008), (4, 8192, 28672), (4, 4096, 11008), (4, 4096 + 16, 14336)):        # (the last two: one fragment each / 4 x 2 units)
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True),
                            enable_tuning=False)
        assert not mm.plans[M]["name"].endswith("xdlt"), mm.plans[M]["name"]


@pytest.mark.parametrize("M", [4, 16])
def test_even_group_count_takes_the_counted_form(M, monkeypatch):
    """K / g = 30: the hand-counted persistent form with 4-byte aligned metadata loads"""
    case = make_case(M, 11008, 3840, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.05, seed=M)
    _both(case, M, monkeypatch)
