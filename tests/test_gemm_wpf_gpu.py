"""The wave-per-fragment form of the one-launch decode member (csrc/wqaa_gemm_kernel.h member 213, plan suffix `xdlw`; round 5): decode
batches M = 3 ... 16 on wide outputs where the whole activation tile fits LDS (K <= 4096 at M <= 16, K <= 8192 at M <= 8) - 4-bit
weights x float16, Scale (+ Zeros) per 128 (the reference's dequantize GEMM at these row counts:
ops/general_matmul/tilelang/dequantize/matmul_dequantize.py:93-109).

The workgroup's eight waves share the activation tile; each adds ALL of K for its own 16-row weight fragments in one accumulator and
stores them itself: no meeting, no partial sums, one launch.  The order of the sum differs from the forms whose waves split K - so the
checks are the oracle's contract, run-to-run bit identity, and hipGraph replays."""
import numpy as np
import pytest

from helpers import set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


def _bits(x):
    return x.view(np.uint16) if x.dtype == np.float16 else x.view(np.uint32)


def _run(case, M, monkeypatch):
    set_knobs(monkeypatch, "gemm", decode_long="4")        # the form wherever it fits
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlw"), mm.plans[M]["name"]
    assert mm.plans[M]["split_k"] == 1 and mm.lib.workspace_bytes(M) == 0
    assert_fp_parity(got, oracle_output(case))
    again, _ = hip_output(case, matmul=mm)
    assert np.array_equal(_bits(got), _bits(again)), "run to run"
    return got, mm


@pytest.mark.parametrize("M", [3, 8, 13, 16])
def test_uint4_scale_zeros_wide_output(M, monkeypatch):
    case = make_case(M, 5504, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=M)
    _run(case, M, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["rescale", "original"])
def test_ragged_n_bias_and_k_8192(zeros_mode, monkeypatch):
    """N = 1000 (63 fragments, the last one 8 rows), bias, K = 8192 at M = 5 (64 k-steps x 2 KiB of tile)"""
    case = make_case(5, 1000, 8192, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode, with_bias=True,
                     scale_mul=0.02, seed=9)
    _run(case, 5, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["original", "rescale"])
def test_uint4_zero_points_in_the_plain_checkpoint_layout(zeros_mode, monkeypatch):
    """`fast_decoding=False` checkpoints (general_compress order, no LOP3 interleave) with zero points: their own instantiations"""
    case = make_case(7, 1536, 4096, W_dtype="uint4", fast_decoding=False, group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode,
                     scale_mul=0.02, seed=17)
    _run(case, 7, monkeypatch)


@pytest.mark.parametrize("fast", [False, True])
def test_int4_scale_only_both_checkpoint_layouts(fast, monkeypatch):
    case = make_case(12, 2048, 4096, W_dtype="int4", fast_decoding=fast, group_size=128, with_scaling=True, scale_mul=0.02, seed=3)
    _run(case, 12, monkeypatch)


def test_nf4_and_float32_output(monkeypatch):
    case = make_case(7, 1536, 2048, W_dtype="nf4", group_size=128, with_scaling=True, scale_mul=0.05, seed=5)
    _run(case, 7, monkeypatch)
    case = make_case(9, 768, 4352, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", out_dtype="float32",
                     scale_mul=0.02, seed=11)                      # (34 k-steps: the last block is half empty)
    _run(case, 9, monkeypatch)


def test_where_it_does_not_fit_other_members_run(monkeypatch):
    """M = 16 at K = 8192 needs 256 KiB of tile; g = 64 is not a hand-counted format"""
    set_knobs(monkeypatch, "gemm", decode_long="4")
    for kw in (dict(M=16, K=8192, g=128), dict(M=8, K=4096, g=64)):
        case = make_case(kw["M"], 512, kw["K"], W_dtype="uint4", group_size=kw["g"], with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=2)
        got, mm = hip_output(case)
        assert not mm.plans[kw["M"]]["name"].endswith("xdlw")
        assert_fp_parity(got, oracle_output(case))


def test_hipgraph_replays(monkeypatch):
    import torch
    set_knobs(monkeypatch, "gemm", decode_long="4")
    M = 8
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=21)
    ref, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlw")
    dev = "cuda"
    A = torch.from_numpy(case["A"]).to(dev)
    qw = mm.transform_weight(torch.from_numpy(case["codes"])).to(dev)
    sc = torch.from_numpy(case["scale"]).to(dev)
    zr = torch.from_numpy(case["zeros"]).to(dev)
    out = torch.zeros((M, 4096), dtype=torch.float16, device=dev)
    g = torch.cuda.CUDAGraph()                                    # (no scratch: capture needs no earlier call)
    with torch.cuda.graph(g):
        mm.forward(A, qw, scale=sc, zeros=zr, output=out)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))
