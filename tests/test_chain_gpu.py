"""wqaa_matmul_chain (include/wqaa.h; csrc/wqaa_chain.hip): a chain of dependent operators - the post-attention half of a
decoder layer, o_proj (+ residual) -> RMSNorm -> gate / up * silu -> down_proj (+ residual), the reference's
integration/BitNet/modeling_bitnet.py:240-244, :839-860 - described once and run as the launches it stands for
(`Matmul.forward_ex`, `matmul_gate_up`).

Every stage's output is checked bit for bit against those launches made one by one on the same inputs (a whole-chain
tolerance would hide a wrong sub-stage), over formats, ragged shapes, repeated calls, a captured hipGraph, temporaries the
caller never sees (`output=False`), and against the oracle's restatement of the layer at the exact-product members'
tolerance.  (The persistent one-launch member rounds 3-5 tested here is gone: csrc/wqaa_chain.hip's header.)"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from bitblas_amd.chain import ChainStep, chain_plan, matmul_chain
from helpers import _to_dev, assert_fp_parity, make_case

pytestmark = pytest.mark.gpu


def build(case):
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    w = (W, _to_dev(case["scale"], "cuda"), _to_dev(case["zeros"], "cuda"), _to_dev(case["bias"], "cuda"))
    return mm, w


def layer(H, I, wd="int4", g=128, zm=None, bias=False, seed=0, scale_mul=0.03):
    kw = dict(W_dtype=wd, group_size=g, with_scaling=True, with_zeros=zm is not None, zeros_mode=zm or "original", with_bias=bias,
              scale_mul=scale_mul)
    cases = dict(o=make_case(1, H, H, seed=seed + 1, **kw), gate=make_case(1, I, H, seed=seed + 2, **kw),
                 up=make_case(1, I, H, seed=seed + 3, **kw), down=make_case(1, H, I, seed=seed + 4, **kw))
    ops = {k: build(c) for k, c in cases.items()}
    rng = np.random.default_rng(seed)
    attn = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    x = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    nw = torch.from_numpy((1.0 + (rng.random(H, dtype=np.float32) - 0.5) * 0.2).astype(np.float16)).cuda()
    return cases, ops, attn, x, nw


def by_launches(ops, attn, x, nw, eps):
    (o, wo), (g, wg), (u, wu), (d, wd) = ops["o"], ops["gate"], ops["up"], ops["down"]
    h = o.forward_ex(attn, wo[0], scale=wo[1], zeros=wo[2], bias=wo[3], residual=x)
    act = bitblas.matmul_gate_up(g, u, h, wg, wu, norm=(nw, eps))
    out = d.forward_ex(act, wd[0], scale=wd[1], zeros=wd[2], bias=wd[3], residual=h)
    return h, act, out


def tail_steps(ops, attn, x, nw, eps, keep=True):
    (o, wo), (g, wg), (u, wu), (d, wd) = ops["o"], ops["gate"], ops["up"], ops["down"]
    return [ChainStep(o, wo, attn, residual=x, output=None if keep else False),
            ChainStep(g, wg, 0, norm=(nw, eps), up_op=u, up_weights=wu, output=None if keep else False),
            ChainStep(d, wd, 1, residual=0)]


LAYERS = [  # (hidden, intermediate, W_dtype, group, zeros_mode, bias)
    (4096, 11008, "int4", 128, None, False),          # Llama-2-7B, BASELINE c2's format
    (1024, 2048, "int4", 128, None, False),
    (4096, 9728, "uint4", 128, "original", True),     # K = 9728: a ragged last lane chunk (4.75), zeros, bias
    (1536, 4096, "uint4", 64, "rescale", False),
    (1024, 3072, "int2", 128, None, True),
    (1056, 2048, "int4", -1, None, False),            # 528 tasks over 256 CUs: uneven ranges; one group per row
]


@pytest.mark.parametrize("H,I,wd,g,zm,bias", LAYERS)
def test_decoder_tail_bit_for_bit(H, I, wd, g, zm, bias):
    if H % (128 // int(wd[-1])) or I % (128 // int(wd[-1])):
        pytest.skip("K must be a multiple of the lane chunk")
    cases, ops, attn, x, nw = layer(H, I, wd, g, zm, bias, seed=H + I)
    eps = 1e-5
    steps = tail_steps(ops, attn, x, nw, eps)
    plan = chain_plan(steps)
    assert plan["launches"] == 3 and plan["plan"] is None, plan
    want = by_launches(ops, attn, x, nw, eps)
    got = matmul_chain(steps)
    torch.cuda.synchronize()
    for name, a, b in zip(("o_proj + x", "silu(gate) * up", "down_proj + h"), got, want):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} elements differ from the launch's"
    # intermediate outputs not stored: the same final bits
    got2 = matmul_chain(tail_steps(ops, attn, x, nw, eps, keep=False))
    torch.cuda.synchronize()
    assert got2[0] is None and got2[1] is None and torch.equal(got2[2], want[2])


def test_decoder_tail_against_the_oracle():
    H, I = 1024, 2048
    cases, ops, attn, x, nw = layer(H, I, seed=5)
    eps = 1e-5
    got = matmul_chain(tail_steps(ops, attn, x, nw, eps))
    torch.cuda.synchronize()

    def exact(c, A):
        return oracle.matmul_dequant_exact(A, c["codes"], source_format=c["source_format"], bit=c["bit"], scale=c["scale"], zeros=c["zeros"],
                                           zeros_mode=c["zeros_mode"], group_size=c["g"], bias=c["bias"], out_dtype="float16")
    h = oracle.add_residual_f16(exact(cases["o"], attn.cpu().numpy()), x.cpu().numpy())
    assert_fp_parity(got[0].cpu().numpy(), h, rtol=1e-3, atol_frac=6e-4)
    hn = oracle.rms_norm_f16(got[0].cpu().numpy(), nw.cpu().numpy(), eps)
    act = oracle.silu_mul_f16(exact(cases["gate"], hn), exact(cases["up"], hn))
    assert_fp_parity(got[1].cpu().numpy(), act.astype(np.float32), rtol=4e-3, atol_frac=2e-3)
    out = oracle.add_residual_f16(exact(cases["down"], got[1].cpu().numpy()), got[0].cpu().numpy())
    assert_fp_parity(got[2].cpu().numpy(), out, rtol=1e-3, atol_frac=6e-4)


def test_repeated_launches_and_graph_replay():
    cases, ops, attn, x, nw = layer(1024, 2048, seed=11)
    eps = 1e-5
    want = by_launches(ops, attn, x, nw, eps)
    steps = tail_steps(ops, attn, x, nw, eps)
    out = torch.empty_like(want[2])
    steps[2].output = out
    for _ in range(3):
        out.zero_()
        matmul_chain(steps)
        torch.cuda.synchronize()
        assert torch.equal(out, want[2])
    # under capture every buffer must be the caller's (a temporary allocated inside the capture would not outlive it)
    h_buf, a_buf = torch.empty_like(want[0]), torch.empty_like(want[1])
    steps[0].output, steps[1].output = h_buf, a_buf
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        matmul_chain(steps)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            matmul_chain(steps)
            matmul_chain(steps)                  # two chains back to back in one graph
        for _ in range(4):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want[2]) and torch.equal(h_buf, want[0]) and torch.equal(a_buf, want[1])
    # new inputs through the same chain
    x2 = (x * 0.5).contiguous()
    want2 = by_launches(ops, attn, x2, nw, eps)
    got2 = matmul_chain(tail_steps(ops, attn, x2, nw, eps))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got2, want2))


def test_shared_input_and_longer_chains():
    """q / k / v-like steps that read ONE vector through one norm; a chain may continue into the next layer's projections"""
    H, I = 1024, 2048
    cases, ops, attn, x, nw = layer(H, I, seed=21)
    eps = 1e-5
    h, act, out = by_launches(ops, attn, x, nw, eps)
    qkv = [build(make_case(1, n, H, W_dtype="int4", group_size=128, with_scaling=True, seed=100 + i, scale_mul=0.03)) for i, n in enumerate((4096, 2048, 2048))]
    nw2 = (nw * 1.1).to(torch.float16).contiguous()
    want_qkv = [op.forward_ex(out, w[0], scale=w[1], norm=(nw2, eps)) for op, w in qkv]
    steps = tail_steps(ops, attn, x, nw, eps) + [ChainStep(op, w, 2, norm=(nw2, eps)) for op, w in qkv]
    plan = chain_plan(steps)
    assert plan["launches"] == 6, plan
    got = matmul_chain(steps)
    torch.cuda.synchronize()
    for a, b in zip(got, [h, act, out] + want_qkv):
        assert torch.equal(a, b)
    # a chain that starts with the norm on the caller's vector, three operators on one staged tile
    steps = [ChainStep(op, w, out, norm=(nw2, eps)) for op, w in qkv]
    got = matmul_chain(steps)
    torch.cuda.synchronize()
    for a, b in zip(got, want_qkv):
        assert torch.equal(a, b)


def test_k_split_stage_and_temporaries():
    """a stage whose single launch splits K across waves, zeros + bias; intermediates as temporaries the caller never sees"""
    cases, ops, attn, x, nw = layer(2048, 5632, "uint4", 128, "original", True, seed=3)
    eps = 1e-5
    steps = tail_steps(ops, attn, x, nw, eps)
    assert chain_plan(steps)["launches"] == 3
    want = by_launches(ops, attn, x, nw, eps)
    got = matmul_chain(steps)
    got2 = matmul_chain(tail_steps(ops, attn, x, nw, eps, keep=False))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want)) and torch.equal(got2[2], want[2])


def test_two_row_chain():
    """m = 2: the same call, the launches' results"""
    H, I = 1024, 2048
    kw = dict(W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.03)
    o = build(make_case(2, H, H, seed=1, **kw))
    d = build(make_case(2, H, H, seed=2, **kw))
    rng = np.random.default_rng(0)
    a = torch.from_numpy((rng.random((2, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    steps = [ChainStep(o[0], o[1], a, residual=a), ChainStep(d[0], d[1], 0, residual=0, output=None)]
    assert chain_plan(steps)["launches"] == 2
    got = matmul_chain(steps)
    h = o[0].forward_ex(a, o[1][0], scale=o[1][1], residual=a)
    want = d[0].forward_ex(h, d[1][0], scale=d[1][1], residual=h)
    torch.cuda.synchronize()
    assert torch.equal(got[0], h) and torch.equal(got[1], want)


def test_decoder_tail_module():
    """`bitblas_amd.DecoderTail` over four `Linear` layers: its three launches and `matmul_chain` over its steps give the same
    bits, and both are the reference's layer (torch's elementwise kernels around the layers' own forwards) within the members'
    tolerance"""
    H, I = 1024, 2048
    rng = np.random.default_rng(7)

    def linear(n_in, n_out):
        lin = bitblas.Linear(n_in, n_out, bias=False, A_dtype="float16", W_dtype="uint4", accum_dtype="float16", out_dtype="float16",
                             group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", opt_M=[1], enable_tuning=False)
        lin.load_and_transform_weight(torch.from_numpy(rng.integers(0, 16, size=(n_out, n_in)).astype(np.int8)),
                                      scales=torch.from_numpy((rng.random((n_out, n_in // 128), dtype=np.float32) * 0.01).astype(np.float16)),
                                      zeros=torch.from_numpy(rng.integers(6, 10, size=(n_out, n_in // 128)).astype(np.float16)))
        return lin.cuda()

    o, gate, up, down = linear(H, H), linear(H, I), linear(H, I), linear(I, H)
    nw = torch.from_numpy((1.0 + (rng.random(H, dtype=np.float32) - 0.5) * 0.2).astype(np.float16)).cuda()
    attn = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    x = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    tail = bitblas.DecoderTail(o, gate, up, down, nw, eps=1e-5)
    assert chain_plan(tail.steps(attn, x))["launches"] == 3
    a, b = tail(attn, x), matmul_chain(tail.steps(attn, x))[2]
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    h = x + o(attn)
    hn = torch.nn.functional.rms_norm(h, (H,), nw, 1e-5)
    want = h + down(torch.nn.functional.silu(gate(hn)) * up(hn))
    # (torch's norm sums x^2 in another order, its layers round where the fused ops do not: a last float16 bit here and there)
    assert_fp_parity(b.cpu().numpy(), want.float().cpu().numpy(), rtol=4e-3, atol_frac=4e-3)


def test_strict_reference_plain_items():
    """a PLAIN item (no norm, no residual) stands for `wqaa_matmul`, which runs a per-element-rounding member for a
    strict_reference operator: the chain gives that launch's bits"""
    H = 4096
    kw = dict(W_dtype="int4", group_size=128, with_scaling=True, scale_mul=0.03)

    def strict(case):
        mm = bitblas.Matmul(case["config"], enable_tuning=False, strict_reference=True)
        W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
        return mm, (W, _to_dev(case["scale"], "cuda"), None, None)
    a_op = strict(make_case(1, H, H, seed=1, **kw))
    b_op = strict(make_case(1, H, H, seed=2, **kw))
    rng = np.random.default_rng(0)
    a = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    steps = [ChainStep(a_op[0], a_op[1], a), ChainStep(b_op[0], b_op[1], 0, output=None)]
    plan = chain_plan(steps)
    assert plan["launches"] == 2, plan
    got = matmul_chain(steps)
    h = a_op[0].forward(a, a_op[1][0], scale=a_op[1][1])
    want = b_op[0].forward(h, b_op[1][0], scale=b_op[1][1])
    torch.cuda.synchronize()
    assert torch.equal(got[0], h) and torch.equal(got[1], want)
