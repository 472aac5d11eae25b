"""wqaa_matmul_group on the GPU: a group of operators gives, BIT FOR BIT, what calling each operator in turn gives
(same kernels, same tile configuration per row, same summation order), whether the library fuses the group into one
launch or not - and the fused launches agree with the oracle like the single ones do.

The projections grouped here are the ones the reference's own integration fuses (by concatenating weights):
integration/BitNet/modeling_bitnet.py `BitnetAttentionQKVFused` :440-585, `BitnetMLPFuseGateUp` :247-290.
"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
from bitblas_amd import group as wgroup
from helpers import set_knobs, case_contract, contract, _to_dev, assert_fp_parity, make_case, oracle_output

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(case, strict):
    """operator + device operands of a helpers.make_case case"""
    mm = bitblas.Matmul(case["config"], enable_tuning=False, strict_reference=strict)
    cfg = case["config"]
    w_user = case["w_user"]
    wt = w_user if isinstance(w_user, torch.Tensor) else torch.from_numpy(w_user)
    if case["source_format"] == "int" and case["bit"] < 8 and (case["bit"] == 1 or cfg.with_scaling or cfg.with_zeros):
        W = mm.weight_transform(torch.from_numpy(case["codes"])).to(DEV)
    else:
        W = mm.transform_weight(wt.to(DEV))
    return mm, (W, _to_dev(case["scale"], DEV), _to_dev(case["zeros"], DEV), _to_dev(case["bias"], DEV))


def run_both(cases, strict, shared_a=True, expect_launches=1):
    ops, weights = zip(*[build(c, strict) for c in cases])
    M = cases[0]["M"]
    if shared_a:
        A = _to_dev(cases[0]["A"], DEV)
        As = [A] * len(cases)
    else:
        As = [_to_dev(c["A"], DEV) for c in cases]
    plan = wgroup.group_plan(ops, M)
    assert plan["launches"] == expect_launches, plan
    single = [op(a, *w) for op, a, w in zip(ops, As, weights)]
    grouped = bitblas.matmul_group(ops, As[0] if shared_a else As, weights)
    torch.cuda.synchronize()
    for i, (s, g) in enumerate(zip(single, grouped)):
        assert g.shape == s.shape and g.dtype == s.dtype
        assert torch.equal(s, g), f"member {i}: grouped launch differs from the single call (max abs {(s.float() - g.float()).abs().max().item():.3g})"
    return ops, grouped, plan


def int4_case(N, K=4096, M=1, seed=0, **kw):
    kw.setdefault("group_size", 128)
    kw.setdefault("with_scaling", True)
    return make_case(M, N, K, W_dtype=kw.pop("W_dtype", "int4"), scale_mul=0.05, seed=seed, **kw)


@pytest.mark.parametrize("strict", [False, True])
def test_qkv_and_gate_up_of_a_llama2_7b_layer(strict):
    """the headline workload's groups at full size: q/k/v 3 x (4096 x 4096), gate/up 2 x (11008 x 4096), M = 1"""
    qkv = [int4_case(4096, seed=s) for s in (1, 2, 3)]
    for c in qkv[1:]:
        c["A"] = qkv[0]["A"]
    ops, outs, plan = run_both(qkv, strict)
    assert plan["plan"]["name"].endswith("_x3")
    for c, o in zip(qkv, outs):
        assert_fp_parity(o.cpu().numpy(), oracle_output(c), **case_contract(c, default_members=not strict, m=1))
    gu = [int4_case(11008, seed=s) for s in (4, 5)]
    gu[1]["A"] = gu[0]["A"]
    ops, outs, plan = run_both(gu, strict)
    # 2 x 688 row-group blocks, one workgroup each (the grid of a group is not capped at what the chip holds at once)
    for c, o in zip(gu, outs):
        assert_fp_parity(o.cpu().numpy(), oracle_output(c), **case_contract(c, default_members=not strict, m=1))


@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("M", [1, 2])
def test_unequal_members_grouped_query_attention(strict, M):
    """k / v narrower than q (and a ragged row count): every member keeps its own N, pointers and tail"""
    cases = [int4_case(N, K=2048, M=M, seed=N, W_dtype="uint4", with_zeros=True, zeros_mode="original") for N in (2048, 512, 272)]
    for c in cases[1:]:
        c["A"] = cases[0]["A"]
    ops, outs, _ = run_both(cases, strict)
    for c, o in zip(cases, outs):
        assert_fp_parity(o.cpu().numpy(), oracle_output(c), **case_contract(c, default_members=not strict, m=1))


@pytest.mark.parametrize("kw", [
    dict(W_dtype="uint4", with_zeros=True, zeros_mode="rescale"),
    dict(W_dtype="uint4", with_zeros=True, zeros_mode="quantized"),
    dict(W_dtype="uint2", with_zeros=True, zeros_mode="original"),
    dict(W_dtype="int1", with_scaling=False, group_size=-1),
    dict(W_dtype="nf4"),
    dict(W_dtype="int4", with_bias=True, fast_decoding=False),
    dict(W_dtype="e4m3_float8", with_scaling=False, group_size=-1),
], ids=lambda kw: kw["W_dtype"] + "_" + str(kw.get("zeros_mode", "")) + ("_bias" if kw.get("with_bias") else ""))
@pytest.mark.parametrize("strict", [False, True])
def test_formats_and_modes(kw, strict):
    cases = [int4_case(N, K=1024, seed=7 + N, **dict(kw)) for N in (1024, 512)]
    cases[1]["A"] = cases[0]["A"]
    run_both(cases, strict)


def test_int2_int8_bitnet_members_bit_exact():
    """W_int2 x A_int8 (BASELINE c4) q/k/v group: integer accumulation, equal to the oracle exactly"""
    cases = [make_case(1, N, 4096, W_dtype="int2", A_dtype="int8", out_dtype="int32", seed=N) for N in (4096, 4096, 1024)]
    for c in cases[1:]:
        c["A"] = cases[0]["A"]
    ops, outs, _ = run_both(cases, True)
    for c, o in zip(cases, outs):
        np.testing.assert_array_equal(o.cpu().numpy(), oracle_output(c))


def test_dense_fp8_members():
    """e4m3 x e4m3 (BASELINE c5) at M = 1: the per-rank q/k/v slices of a column-sharded layer as one launch"""
    def case(N, seed):
        g = torch.Generator().manual_seed(seed)
        cfg = bitblas.MatmulConfig(M=1, N=N, K=8192, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32",
                                   out_dtype="float16")
        mm = bitblas.Matmul(cfg, enable_tuning=False)
        W = (torch.rand((N, 8192), generator=g) * 2 - 1).to(torch.float8_e4m3fn).to(DEV)
        return mm, W
    A = (torch.rand((1, 8192)) * 2 - 1).to(torch.float8_e4m3fn).to(DEV)
    ops, Ws = zip(*[case(N, N) for N in (1024, 128, 128)])
    assert wgroup.group_plan(ops, 1)["launches"] == 1
    single = [op(A, W) for op, W in zip(ops, Ws)]
    grouped = bitblas.matmul_group(ops, A, list(Ws))
    torch.cuda.synchronize()
    for s, g, W in zip(single, grouped, Ws):
        assert torch.equal(s, g)
        want = (A.float() @ W.float().T).cpu().numpy()
        assert_fp_parity(g.cpu().numpy(), want, rtol=1e-3, atol_frac=1e-3)


def test_members_with_their_own_activations():
    cases = [int4_case(1024, K=2048, seed=s) for s in (11, 12)]
    run_both(cases, False, shared_a=False)
    run_both(cases, True, shared_a=False)


def test_unfusable_groups_run_member_by_member():
    cases = [int4_case(1024, K=2048, seed=1), int4_case(1024, K=1024, seed=2)]        # different K
    run_both(cases, False, shared_a=False, expect_launches=2)
    cases = [int4_case(512, K=1024, M=16, seed=s) for s in (3, 4)]                     # MFMA members (M = 16)
    cases[1]["A"] = cases[0]["A"]
    run_both(cases, True, expect_launches=2)
    cases = [int4_case(512, K=4096, M=128, seed=s, W_dtype="uint4", with_zeros=True) for s in (5, 6)]   # split-K members: workspace
    cases[1]["A"] = cases[0]["A"]
    run_both(cases, True, expect_launches=2)


def test_many_row_single_launch_on_a_capped_grid_matches_oracle():
    """N = 22016 (a concatenated gate/up): more row-group blocks than workgroups the chip holds at once - workgroups
    iterate over their XCD's eighth of the blocks (xcd_row_blocks).  Exact and rounding members against the oracle."""
    c = int4_case(22016, seed=9)
    for strict in (False, True):
        mm, w = build(c, strict)
        plan = mm.plans[1]
        out = mm(_to_dev(c["A"], DEV), *w)
        torch.cuda.synchronize()
        assert_fp_parity(out.cpu().numpy(), oracle_output(c), rtol=1e-3, atol_frac=1.5e-3)
        assert plan["grid"] >= 256


def test_results_do_not_depend_on_the_grid(monkeypatch):
    """fewer workgroups than row-group blocks (workgroups take several blocks of their XCD's eighth, csrc/wqaa_kinds.h
    xcd_row_blocks), single launches of both families and a group launch: bit-identical to the one-block-per-workgroup grids"""
    c = int4_case(11008, seed=21)
    cases = [c, dict(int4_case(11008, seed=22), A=c["A"])]
    A = _to_dev(c["A"], DEV)
    for strict in (False, True):
        built = [build(x, strict) for x in cases]
        ops = [b[0] for b in built]
        ws = [b[1] for b in built]
        base = [op(A, *w) for op, w in zip(ops, ws)]
        gbase = bitblas.matmul_group(ops, A, ws)
        for grid, ggrid in ((344, 344), (504, 200), (8, 16)):
            set_knobs(monkeypatch, "gemv", grid=str(grid))
            set_knobs(monkeypatch, "gemv", grid=str(grid))
            set_knobs(monkeypatch, "gemv", group_grid=str(ggrid))
            plan = ops[0].lib.plan(1)                     # planning re-reads the tuning variables
            assert plan["grid"] == grid
            gplan = wgroup.group_plan(ops, 1)
            assert gplan["plan"]["grid"] == 2 * ggrid
            for op, w, b in zip(ops, ws, base):
                assert torch.equal(op(A, *w), b)
            for g_, b in zip(bitblas.matmul_group(ops, A, ws), gbase):
                assert torch.equal(g_, b)
        set_knobs(monkeypatch, "gemv", grid=None)
        set_knobs(monkeypatch, "gemv", grid=None)
        set_knobs(monkeypatch, "gemv", group_grid=None)
        ops[0].lib.plan(1)
        for g_, b in zip(gbase, base):
            assert torch.equal(g_, b)
    torch.cuda.synchronize()


def test_linear_group_and_graph_replay():
    """`q, k, v = LinearGroup([q_proj, k_proj, v_proj])(x)`: equal to the layers' own forward; parameter swaps are
    seen; the call captures into a hipGraph and replays on new activations."""
    torch.manual_seed(0)
    layers = []
    for N in (1024, 256, 256):
        lin = bitblas.Linear(1024, N, bias=False, A_dtype="float16", W_dtype="uint4", accum_dtype="float16", out_dtype="float16",
                             group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", opt_M=[1, 16], enable_tuning=False)
        lin = lin.to(DEV)
        w = torch.randint(0, 16, (N, 1024), dtype=torch.int8)
        lin.load_and_transform_weight(w.to(DEV), scales=(torch.rand(N, 8) * 0.05).half().to(DEV),
                                      zeros=torch.full((N, 8), 8.0).half().to(DEV))
        layers.append(lin)
    grp = bitblas.LinearGroup(layers).to(DEV)
    x = (torch.rand(1, 1024, device=DEV) - 0.5).half()
    want = [l(x) for l in layers]
    got = grp(x)
    torch.cuda.synchronize()
    for w_, g_ in zip(want, got):
        assert torch.equal(w_, g_)
    # M = 16: member by member (MFMA family), same results
    x16 = (torch.rand(16, 1024, device=DEV) - 0.5).half()
    for w_, g_ in zip([l(x16) for l in layers], grp(x16)):
        assert torch.equal(w_, g_)
    # a parameter replaced after the first call must be picked up
    layers[1].scales = (torch.rand(256, 8, device=DEV) * 0.05).half()
    want1 = layers[1](x)
    assert torch.equal(grp(x)[1], want1)
    # capture + replay
    outs = [torch.empty_like(w_) for w_ in want]
    xs = x.clone()
    grp(xs, outputs=outs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        grp(xs, outputs=outs)
    x2 = (torch.rand(1, 1024, device=DEV) - 0.5).half()
    xs.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    for l, o in zip(layers, outs):
        assert torch.equal(l(x2), o)


def test_groups_whose_members_alone_would_run_differently_stay_bit_identical():
    """The tile configuration of a fused launch is chosen for the MERGED row count.  Where a member alone would land on
    another family (M = 2, N < 8192: rounding members; the merged 3 x 4096 would take exact products) or another K split
    across waves (few rows x long K), fusing would change its bits - those groups run member by member, and either way
    the result is the single calls', bit for bit."""
    qkv = [int4_case(4096, M=2, seed=s) for s in (31, 32, 33)]
    for c in qkv[1:]:
        c["A"] = qkv[0]["A"]
    ops = [build(c, False)[0] for c in qkv]
    launches = wgroup.group_plan(ops, 2)["launches"]
    run_both(qkv, False, expect_launches=launches)
    # long K, few rows: a member of 1024 rows splits K four ways alone, the merged 2048 rows would not
    longk = [int4_case(1024, K=11008, M=1, seed=s) for s in (41, 42)]
    longk[1]["A"] = longk[0]["A"]
    ops = [build(c, False)[0] for c in longk]
    alone, merged = ops[0].plans[1]["split_k"], wgroup.group_plan(ops, 1)
    if merged["launches"] == 1:
        assert merged["plan"]["split_k"] == alone, (alone, merged)
    run_both(longk, False, expect_launches=merged["launches"])
    for strict in (False, True):
        longk8 = [int4_case(768, K=8192, M=1, seed=s) for s in (51, 52, 53)]
        for c in longk8[1:]:
            c["A"] = longk8[0]["A"]
        ops = [build(c, strict)[0] for c in longk8]
        run_both(longk8, strict, expect_launches=wgroup.group_plan(ops, 1)["launches"])


def test_same_total_other_split_do_not_share_a_verdict():
    """two groups of one total N and member count but different splits, one after the other in one process (the
    fusability memo was keyed by the total only): each must give its members the single calls' bits, whatever the other group's
    verdict was.  Few-row members (1024 x 8192: K split across waves alone) next to many-row ones make the splits differ."""
    K = 8192
    for Ns in ((4096, 4096, 4096), (10240, 1024, 1024), (4096, 4096, 4096), (8192, 2048, 2048)):
        cases = [int4_case(n, K=K, seed=n + i) for i, n in enumerate(Ns)]
        for c in cases[1:]:
            c["A"] = cases[0]["A"]
        ops, weights = zip(*[build(c, False) for c in cases])
        A = _to_dev(cases[0]["A"], DEV)
        single = [op(A, *w) for op, w in zip(ops, weights)]
        grouped = bitblas.matmul_group(ops, A, weights)
        torch.cuda.synchronize()
        for i, (s_, g_) in enumerate(zip(single, grouped)):
            assert torch.equal(s_, g_), f"split {Ns}, member {i}: grouped launch differs from the single call"
