"""The mid-M member (csrc/wqaa_gemm_mid_kernel.h, plan suffix `xmk`; round 5): W int4 / uint4 x A float16 at M = 17 ... 128, K in 8
slices whose partial sums a small second launch adds in slice order - BASELINE c3's M = 128 config (`W_int4 A_fp16 GEMM, M in
{16, 128, 4096}, N = K = 4096, group_size = 128 with zeros`) and the M = 32 / 64 steps of the reference's default `opt_M` list
(ops/general_matmul/__init__.py:188-192; the split-K heuristic it replaces: tilelang/dequantize/matmul_dequantize_mma.py:127-168).

Checked against the CPU oracle, bit for bit against itself run to run, under hipGraph replays and with launches in flight on two
streams (each stream's scratch holds its own partial sums), and against the two-launch member it stands in for (oracle
tolerance: the slices are other k ranges there).  (Round 5's in-launch meeting of the slices - slower, opt-in - and its three
reduction paths were removed in round 6 together with their tests.)"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
from helpers import set_knobs, assert_fp_parity, hip_output, make_case, oracle_output

pytestmark = pytest.mark.gpu


def _bits(x):
    return x.view(np.uint16) if x.dtype == np.float16 else x.view(np.uint32)


def _run(case, M, monkeypatch, check_paths=True):
    """the member against the oracle, and run to run bit for bit"""
    set_knobs(monkeypatch, "gemm", mid=None)
    set_knobs(monkeypatch, "gemm", mid=2)         # (every shape the member takes, not only where it measured ahead)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xmk"), mm.plans[M]["name"]
    assert mm.plans[M]["split_k"] == 8
    assert_fp_parity(got, oracle_output(case))
    again, _ = hip_output(case, matmul=mm)
    assert np.array_equal(_bits(got), _bits(again)), "run to run"
    return got, mm


@pytest.mark.parametrize("M", [17, 32, 33, 64, 100, 128])
def test_c3_uint4_scale_zeros_4096(M, monkeypatch):
    """BASELINE c3 at N = K = 4096 (uint4, g = 128, zeros original) - every tile height, ragged M included"""
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=M)
    _run(case, M, monkeypatch)


@pytest.mark.parametrize("zeros_mode", ["rescale", "quantized"])
@pytest.mark.parametrize("M", [48, 128])
def test_other_zero_point_forms_and_bias(zeros_mode, M, monkeypatch):
    case = make_case(M, 2048 + 128, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode=zeros_mode, with_bias=True,
                     scale_mul=0.02, seed=7 + M)
    _run(case, M, monkeypatch)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("cfg", [dict(with_scaling=True, group_size=128), dict()])
def test_int4_both_checkpoint_layouts(cfg, fast, monkeypatch):
    """signed int4, plain and LOP3-interleaved (`fast_decoding`) checkpoint layouts, with the group scale and without any"""
    case = make_case(96, 1024, 4096, W_dtype="int4", fast_decoding=fast, scale_mul=0.02, seed=3, **cfg)
    _run(case, 96, monkeypatch, check_paths=fast)


def test_group_sizes_the_wide_metadata_loads_do_not_take_keep_their_members(monkeypatch):
    set_knobs(monkeypatch, "gemm", mid=2)
    for g in (-1, 32, 256):             # (one load fetches Scale / Zeros of a wave's consecutive k-steps: one group per k-step, g = 128)
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=96, N=1024, K=4096, A_dtype="float16", W_dtype="int4", group_size=g, with_scaling=True), enable_tuning=False)
        assert "xmk" not in mm.plans[96]["name"], mm.plans[96]["name"]


@pytest.mark.parametrize("M,N,K", [(64, 2048, 8192), (32, 4096, 8192), (40, 1000, 4096), (128, 132, 4096)])
def test_other_k_and_ragged_n(M, N, K, monkeypatch):
    """K = 8192 (four k-steps per k-half: M <= 64 fits the LDS), N off the 128-column tile"""
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=M + N)
    _run(case, M, monkeypatch)


def test_float32_output(monkeypatch):
    case = make_case(128, 2048, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, out_dtype="float32", scale_mul=0.02, seed=5)
    _run(case, 128, monkeypatch)


def test_hipgraph_replays(monkeypatch):
    """one captured pair of launches replayed: same kernel arguments, same scratch every time"""
    set_knobs(monkeypatch, "gemm", mid=None)
    set_knobs(monkeypatch, "gemm", mid=2)
    M = 128
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.02, seed=9)
    ref, mm = hip_output(case)
    dev = "cuda"
    A = torch.from_numpy(case["A"]).to(dev)
    qw = mm.transform_weight(torch.from_numpy(case["codes"])).to(dev)
    sc = torch.from_numpy(case["scale"]).to(dev)
    zr = torch.from_numpy(case["zeros"]).to(dev)
    out = torch.zeros((M, 4096), dtype=torch.float16, device=dev)
    mm.forward(A, qw, scale=sc, zeros=zr, output=out)          # (outside capture first: the stream's scratch exists)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mm.forward(A, qw, scale=sc, zeros=zr, output=out)
        mm.forward(A, qw, scale=sc, zeros=zr, output=out)      # (two launches on one workspace, back to back)
    for _ in range(5):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))


def test_two_streams_do_not_share_partial_sums(monkeypatch):
    """two operators' launches in flight on two streams: each stream's scratch holds its own slices"""
    set_knobs(monkeypatch, "gemm", mid=None)
    set_knobs(monkeypatch, "gemm", mid=2)
    M = 64
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.02, seed=21)
    ref, mm = hip_output(case)
    dev = "cuda"
    A = torch.from_numpy(case["A"]).to(dev)
    qw = mm.transform_weight(torch.from_numpy(case["codes"])).to(dev)
    sc = torch.from_numpy(case["scale"]).to(dev)
    zr = torch.from_numpy(case["zeros"]).to(dev)
    outs = [torch.zeros((M, 4096), dtype=torch.float16, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for _ in range(20):
        for s, o in zip(streams, outs):
            with torch.cuda.stream(s):
                mm.forward(A, qw, scale=sc, zeros=zr, output=o)
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(_bits(o.cpu().numpy()), _bits(ref))


def test_the_member_it_stands_in_for_is_still_there(monkeypatch):
    """WQAA_GEMM_TUNE=mid=0: the two-launch member (split-K + reduce kernel); both within the oracle's tolerance"""
    M = 128
    case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, scale_mul=0.02, seed=2)
    set_knobs(monkeypatch, "gemm", mid=None)
    got, mm = hip_output(case)
    assert mm.plans[M]["name"].endswith("xmk")            # (BASELINE c3's M = 128: the selector's own choice)
    set_knobs(monkeypatch, "gemm", mid="0")
    old, mm0 = hip_output(case)
    assert "xmk" not in mm0.plans[M]["name"] and "xr" in mm0.plans[M]["name"], mm0.plans[M]["name"]
    want = oracle_output(case)
    assert_fp_parity(got, want)
    assert_fp_parity(old, want)


def test_where_the_selector_takes_the_member(monkeypatch):
    """the measured rule (csrc/wqaa_gemm.hip, profiles/r05_ab_mid_v3.txt): 65 ... 128 rows in one round at K = 4096; up to 64 rows on long
    K (8192) or over several rounds of workgroups; never at M <= 16, K off the 2048 grid, other formats, 128 rows over several rounds"""
    set_knobs(monkeypatch, "gemm", mid=None)

    def name(M, N, K, **kw):
        cfg = dict(A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True)
        cfg.update(kw)
        return bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, **cfg), enable_tuning=False).plans[M]["name"]
    for (M, N, K) in ((128, 4096, 4096), (96, 4096, 4096), (65, 3072, 4096), (64, 4096, 8192), (32, 4096, 8192), (64, 8192, 4096), (64, 11008, 4096), (17, 8192, 8192)):
        assert name(M, N, K).endswith("xmk"), (M, N, K, name(M, N, K))
    for (M, N, K) in ((128, 11008, 4096), (128, 4096, 11008), (16, 4096, 4096), (64, 4096, 4096), (32, 4096, 4096), (128, 4096, 2048), (128, 4096, 8192),
                      (256, 4096, 4096), (128, 2048, 4096)):
        assert "xmk" not in name(M, N, K), (M, N, K, name(M, N, K))
    assert "xmk" not in name(128, 4096, 4096, W_dtype="uint2")
