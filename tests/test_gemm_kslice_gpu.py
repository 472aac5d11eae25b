"""The K-sliced form of the one-launch decode member (csrc/wqaa_gemm_kernel.h member 212, plan suffix `xdlk`; round 5): decode batches
M = 3 ... 16 on long K (K > 4096) - 4-bit weights x float16, Scale (+ Zeros) per 128.  The reference runs these row counts through
its split-K-less dequantize GEMM (ops/general_matmul/tilelang/dequantize/matmul_dequantize.py:93-109) whatever K is.

A workgroup owns ONE eighth of K (its activations staged once, shared by its waves) and walks weight fragments; the eight slices'
fp32 partial sums are added by a second launch in slice order - the order the one-launch forms' waves meet in, so the results must
be the same BITS as those forms' (`xdl`, `xdlt`), and within the contract of the CPU oracle."""
import numpy as np
import pytest

from helpers import set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


def _bits(x):
    return x.view(np.uint16) if x.dtype == np.float16 else x.view(np.uint32)


def _run(case, M, monkeypatch):
    set_knobs(monkeypatch, "gemm", decode_long="3")        # the form wherever it fits, not only where it measured ahead
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlk"), mm.plans[M]["name"]
    assert mm.plans[M]["split_k"] == 8
    assert mm.lib.workspace_bytes(M) >= ((case["N"] + 15) // 16) * 8 * 1024
    assert_fp_parity(got, oracle_output(case))
    again, _ = hip_output(case, matmul=mm)
    assert np.array_equal(_bits(got), _bits(again)), "run to run"
    set_knobs(monkeypatch, "gemm", decode_long="2")        # the round-4 selector
    old, mo = hip_output(case)
    name = mo.plans[M]["name"]
    assert not name.endswith("xdlk"), name
    if name.endswith("xdl") or name.endswith("xdlt"):        # the one-launch forms: same k ranges, same order of the eight partial sums
        assert np.array_equal(_bits(got), _bits(old)), f"differs from {name}"
    else:
        assert_fp_parity(old, oracle_output(case))
    return got, mm


@pytest.mark.parametrize("M", [3, 8, 13, 16])
def test_uint4_scale_zeros_4096x11008(M, monkeypatch):
    """the down projection of a 7B decoder layer: 86 k-steps, the last slice two steps long"""
    case = make_case(M, 4096, 11008, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=M)
    _run(case, M, monkeypatch)


@pytest.mark.parametrize("M", [4, 8, 16])
def test_k_28672(M, monkeypatch):
    """a 70B down projection's K: 28 k-steps per slice, up to 112 KiB of activations per workgroup; N kept small for the oracle's sake"""
    case = make_case(M, 1024, 28672, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=40 + M)
    _run(case, M, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["rescale", "quantized"])
def test_other_zero_point_forms_ragged_n_and_bias(zeros_mode, monkeypatch):
    """N = 1000: 63 fragments, the last one 8 rows; `quantized` zero points are not a hand-counted format - the form must not take it"""
    case = make_case(5, 1000, 8192, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode, with_bias=True,
                     scale_mul=0.02, seed=9)
    if zeros_mode == "quantized":
        set_knobs(monkeypatch, "gemm", decode_long="3")
        got, mm = hip_output(case)
        assert not mm.plans[5]["name"].endswith("xdlk")
        assert_fp_parity(got, oracle_output(case))
    else:
        _run(case, 5, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["original", "rescale"])
def test_uint4_zero_points_in_the_plain_checkpoint_layout(zeros_mode, monkeypatch):
    """`fast_decoding=False` checkpoints (general_compress order, no LOP3 interleave) with zero points: their own instantiations"""
    case = make_case(7, 1536, 8192, W_dtype="uint4", fast_decoding=False, group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode,
                     scale_mul=0.02, seed=17)
    _run(case, 7, monkeypatch)


@pytest.mark.parametrize("fast", [False, True])
def test_int4_scale_only_both_checkpoint_layouts(fast, monkeypatch):
    case = make_case(12, 2048, 8192, W_dtype="int4", fast_decoding=fast, group_size=128, with_scaling=True, scale_mul=0.02, seed=3)
    _run(case, 12, monkeypatch)


def test_nf4_lookup_table(monkeypatch):
    case = make_case(7, 1536, 6144, W_dtype="nf4", group_size=128, with_scaling=True, scale_mul=0.05, seed=5)
    _run(case, 7, monkeypatch)


def test_short_last_slices_and_float32_output(monkeypatch):
    """K = 4352: 34 k-steps in runs of 8 - slice 4 holds two of them, slices 5 ... 7 none (their partial sums are zeros)"""
    case = make_case(9, 768, 4352, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", out_dtype="float32",
                     scale_mul=0.02, seed=11)
    _run(case, 9, monkeypatch)


def test_few_fragments_and_other_group_sizes(monkeypatch):
    """N = 128 (8 fragments: one group of workgroups, one fragment per wave); g = 64 is not a hand-counted format: other members"""
    case = make_case(16, 128, 8192, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=2)
    _run(case, 16, monkeypatch)
    set_knobs(monkeypatch, "gemm", decode_long="3")
    case = make_case(8, 512, 8192, W_dtype="uint4", group_size=64, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=2)
    got, mm = hip_output(case)
    assert not mm.plans[8]["name"].endswith("xdlk")
    assert_fp_parity(got, oracle_output(case))


def test_hipgraph_replays(monkeypatch):
    """captured once (two launches back to back on one buffer of partial sums), replayed"""
    import torch
    set_knobs(monkeypatch, "gemm", decode_long="3")
    M = 8
    case = make_case(M, 2048, 8192, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=21)
    ref, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlk")
    dev = "cuda"
    A = torch.from_numpy(case["A"]).to(dev)
    qw = mm.transform_weight(torch.from_numpy(case["codes"])).to(dev)
    sc = torch.from_numpy(case["scale"]).to(dev)
    zr = torch.from_numpy(case["zeros"]).to(dev)
    out = torch.zeros((M, 2048), dtype=torch.float16, device=dev)
    mm.forward(A, qw, scale=sc, zeros=zr, output=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mm.forward(A, qw, scale=sc, zeros=zr, output=out)
        mm.forward(A, qw, scale=sc, zeros=zr, output=out)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))


def test_caller_workspace_full_and_short(monkeypatch):
    """`wqaa_matmul_opts`: a caller workspace of `wqaa_workspace_bytes` holds the slices' partial sums; one too short for them makes the
    call run the member the form stands in for (a one-launch form here: the same bits) instead of failing"""
    import torch
    set_knobs(monkeypatch, "gemm", decode_long="3")
    M, N, K = 8, 2048, 8192
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=31)
    ref, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xdlk")
    need = mm.lib.workspace_bytes(M)
    assert need >= (N // 16) * 8 * 1024
    dev = "cuda"
    A = torch.from_numpy(case["A"]).to(dev)
    qw = mm.transform_weight(torch.from_numpy(case["codes"])).to(dev)
    sc = torch.from_numpy(case["scale"]).to(dev)
    zr = torch.from_numpy(case["zeros"]).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for nbytes in (need, 4096):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.zeros((M, N), dtype=torch.float16, device=dev)
        mm.lib.run_ws(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), zr.data_ptr(), None, out.data_ptr(), M, stream, ws.data_ptr(), nbytes)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref)), nbytes
