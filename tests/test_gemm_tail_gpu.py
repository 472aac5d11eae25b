"""The remainder of a partial round of 256 x 256 tiles as a second launch of the 128-row tile (csrc/wqaa_gemm.hip gemm_choose /
gemm_launch, GemmArgs::tile_n_off).  A launch costs whole rounds of one workgroup per CU; a shape like
2048 x 11008 (344 tiles = one round + 88) paid a whole tile's latency for the 88.  The selector now hands the last N-tiles to
the 128 x 256 member: two launches, disjoint column bands of one output.  Checked here at sizes the oracle finishes quickly
(the band boundary, ragged M, both accumulator types) and, matrix-wide, bit for bit against the single-launch plan
(`WQAA_GEMM_TUNE=pp_tail=0`) - both tiles add the same products in the same order.
Reference semantics: bitblas/ops/general_matmul/tilelang/dequantize/matmul_dequantize_mma.py:333-508 (one output element =
one k-ordered sum, whatever the tiling)."""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from helpers import set_knobs, assert_fp_parity

pytestmark = pytest.mark.gpu


def _op(M, N, K, monkeypatch, tail, **cfg):
    """an operator planned - and launched: the library plans at the first launch after a select, under the variables of THAT
    moment - with (default) or without the second launch; the caller keeps the environment until it has synchronised"""
    if tail:
        set_knobs(monkeypatch, "gemm", pp_tail=None)
        set_knobs(monkeypatch, "gemm", pp_tile=None)
    else:
        set_knobs(monkeypatch, "gemm", pp_tail="0")
        set_knobs(monkeypatch, "gemm", pp_tile="256")      # the same member over the whole output, one launch
    return bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, **cfg), enable_tuning=False)


@pytest.mark.parametrize("M", [512, 500])
def test_uint4_tail_band_matches_the_single_launch_and_the_oracle(M, monkeypatch):
    N, K, g = 33024, 256, 128                       # 2 x 129 tiles of 256 x 256 = one round + 2: the last N-tile goes to the 128-row member
    cfg = dict(A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True, with_zeros=True, zeros_mode="original")
    rng = np.random.default_rng(M)
    codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    A = (torch.from_numpy(rng.random((M, K), dtype=np.float32)) - 0.5).to(torch.float16)
    scale = (torch.from_numpy(rng.random((N, K // g), dtype=np.float32)) * 0.05).to(torch.float16)
    zeros = torch.from_numpy((8 + rng.integers(-2, 3, size=(N, K // g))).astype(np.float32)).to(torch.float16)
    single = _op(M, N, K, monkeypatch, False, **cfg)
    assert single.plans[M]["name"].endswith("pp"), single.plans[M]["name"]
    W = single.weight_transform(torch.from_numpy(codes)).cuda()
    ref = single(A.cuda(), W, scale=scale.cuda(), zeros=zeros.cuda())
    torch.cuda.synchronize()
    tail = _op(M, N, K, monkeypatch, True, **cfg)
    assert tail.plans[M]["name"].endswith("ppt1"), tail.plans[M]["name"]
    got = tail(A.cuda(), W, scale=scale.cuda(), zeros=zeros.cuda())
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    cols = np.unique(np.concatenate([np.arange(0, 40), np.arange(32768 - 20, 32768 + 20), np.arange(N - 40, N), rng.choice(N, 64, replace=False)]))
    rows = np.unique(np.concatenate([np.arange(0, 4), np.arange(126, 130), np.arange(254, 258), np.arange(M - 4, M)]))
    want = oracle.matmul_dequant(A.float().numpy()[rows], codes[cols], source_format="uint", bit=4, scale=scale.float().numpy()[cols],
                                 zeros=zeros.float().numpy()[cols], zeros_mode="original", group_size=g, out_dtype="float32", strict_reference=False)
    g_ = got[torch.from_numpy(rows).cuda()][:, torch.from_numpy(cols).cuda()].float().cpu().numpy()
    assert_fp_parity(g_, want.astype(np.float16).astype(np.float32), rtol=1e-3, atol_frac=1e-3)


def test_int2_int8_tail_band_is_bit_exact(monkeypatch):
    M, N, K = 512, 33024, 512
    cfg = dict(A_dtype="int8", W_dtype="int2", accum_dtype="int32", out_dtype="int32")
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 4, size=(N, K)).astype(np.int8)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    single = _op(M, N, K, monkeypatch, False, **cfg)
    W = single.weight_transform(torch.from_numpy(codes)).cuda()
    ref = single(torch.from_numpy(A).cuda(), W)
    torch.cuda.synchronize()
    tail = _op(M, N, K, monkeypatch, True, **cfg)
    assert tail.plans[M]["name"].endswith("ppt1"), tail.plans[M]["name"]
    got = tail(torch.from_numpy(A).cuda(), W)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    cols = np.unique(np.concatenate([np.arange(32768 - 8, 32768 + 8), np.arange(N - 8, N), rng.choice(N, 32, replace=False)]))
    want = oracle.matmul_dequant(A[:64], codes[cols], source_format="int", bit=2, a_dtype="int8", out_dtype="int32")
    assert np.array_equal(got[:64][:, torch.from_numpy(cols).cuda()].cpu().numpy(), want)


def test_events_and_graph_replay_cover_both_launches(monkeypatch):
    """a captured graph holds both launches; a second replay over a poisoned output restores every column"""
    M, N, K, g = 512, 33024, 256, 128
    cfg = dict(A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True)
    tail = _op(M, N, K, monkeypatch, True, **cfg)
    assert tail.plans[M]["name"].endswith("ppt1"), tail.plans[M]["name"]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    A = (torch.rand((M, K), device="cuda", generator=gen) - 0.5).to(torch.float16)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device="cuda", generator=gen)
    sc = (torch.rand((N, K // g), device="cuda", generator=gen) * 0.05).to(torch.float16)
    out = torch.empty((M, N), dtype=torch.float16, device="cuda")
    ref = tail(A, W, scale=sc).clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tail(A, W, scale=sc, output=out)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            tail(A, W, scale=sc, output=out)
        out.fill_(float("nan"))
        gr.replay()
    s.synchronize()
    assert torch.equal(out, ref)
