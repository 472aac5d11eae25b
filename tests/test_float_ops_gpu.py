"""The caller's elementwise ops of the float16 decode path folded into the GEMV that produces the vector (include/wqaa.h:
WQAA_EPI_ADD_RESIDUAL, wqaa_matmul_gate_up; `Matmul.forward_ex`, `Linear.forward_ex`, `bitblas_amd.matmul_gate_up`): the
`residual + linear(x)` and `silu(gate_proj(x)) * up_proj(x)` of the reference's decoder layers
(integration/BitNet/modeling_bitnet.py:240-244, :281-287, :839-860), each as ONE launch.

Checked
  * bit for bit against the SAME members' plain results put through torch's own elementwise kernels (`residual + out`,
    `F.silu(g) * u`): the fusion must not show in the bits - for the gated activation where the device's exp / divide round
    like torch's (the test says how many elements differ otherwise, and bounds them by one float16 ulp);
  * against the oracle's restatement (`matmul_dequant_exact` -> `add_residual_f16` / `silu_mul_f16`) at the tolerance of the
    exact-product members (tests/test_gemvx_gpu.py).
"""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from bitblas_amd.lib import WqaaError
from helpers import _to_dev, assert_fp_parity, make_case

pytestmark = pytest.mark.gpu


def build(case, strict_reference=False):
    mm = bitblas.Matmul(case["config"], enable_tuning=False, strict_reference=strict_reference)
    cfg = case["config"]
    if case["source_format"] == "int" and case["bit"] < 8 and (case["bit"] == 1 or cfg.with_scaling or cfg.with_zeros):
        W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    else:
        w = case["w_user"]
        W = mm.transform_weight((w if isinstance(w, torch.Tensor) else torch.from_numpy(w)).cuda())
    args = dict(scale=_to_dev(case["scale"], "cuda"), zeros=_to_dev(case["zeros"], "cuda"), bias=_to_dev(case["bias"], "cuda"))
    return mm, W, args


def exact(case, A):
    return oracle.matmul_dequant_exact(A, case["codes"], source_format=case["source_format"], bit=case["bit"], scale=case["scale"],
                                       zeros=case["zeros"], zeros_mode=case["zeros_mode"], group_size=case["g"], bias=case["bias"],
                                       out_dtype="float16")


SHAPES = [  # (N, K, W_dtype, group, zeros_mode or None) - two-row / one-row members, K split across waves, long K (staging beyond the
    # first round), ragged N
    (4096, 11008, "uint4", 128, "original"),    # Llama-2-7B down_proj
    (4096, 11008, "int4", 128, None),
    (1024, 4096, "uint4", 128, "quantized"),    # K split across the waves
    (11008, 4096, "uint4", 128, "rescale"),     # two rows per wave
    (1000, 2048, "int2", -1, None),
    (4096, 14336, "uint2", 128, "original"),
    (2050, 4096, "uint1", 128, None),
    (515, 1024, "int4", 128, None),
]


GATE_UP = [  # (N, K, W_dtype, group, zeros_mode or None, bias)
    (11008, 4096, "int4", 128, None, False),        # Llama-2-7B gate / up
    (11008, 4096, "uint4", 128, "original", True),
    (2050, 2048, "uint4", 128, "quantized", False),
    (1000, 16384, "uint4", 128, "rescale", False),  # K split across the waves
    (5000, 1024, "int2", -1, None, True),
    (4096, 14336, "uint2", 128, "original", False),
    (515, 1024, "uint1", 128, None, False),
]


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K,wd,g,zm,wb", GATE_UP)
def test_gate_up_pair_one_launch(N, K, wd, g, zm, wb, M):
    kw = dict(W_dtype=wd, group_size=g, with_scaling=True, with_zeros=zm is not None, zeros_mode=zm or "original", with_bias=wb, scale_mul=0.08)
    cg, cu = make_case(M, N, K, seed=N + K + M, **kw), make_case(M, N, K, seed=N + K + M + 7, **kw)
    cu["A"] = cg["A"]
    gate_op, Wg, ag = build(cg)
    up_op, Wu, au = build(cu)
    assert bitblas.gate_up_plan(gate_op, M)["name"].endswith("_pair")
    A = torch.from_numpy(cg["A"]).cuda()
    wg, wu = (Wg, ag["scale"], ag["zeros"], ag["bias"]), (Wu, au["scale"], au["zeros"], au["bias"])
    act = bitblas.matmul_gate_up(gate_op, up_op, A, wg, wu)
    # the two projections through the same family's plain members (a residual of zeros changes no bit), then torch's kernels
    zero = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    g_out = gate_op.forward_ex(A, Wg, residual=zero, **ag)
    u_out = up_op.forward_ex(A, Wu, residual=zero, **au)
    want = torch.nn.functional.silu(g_out) * u_out
    torch.cuda.synchronize()
    got_np, want_np = act.cpu().numpy(), want.cpu().numpy()
    differ = int((got_np.view(np.uint16) != want_np.view(np.uint16)).sum())
    if differ:
        # exp / divide of this kernel and of torch's kernel may round differently in the last fp32 bit: one float16 ulp of silu(g)
        assert differ <= max(2, got_np.size // 200), f"{differ}/{got_np.size} elements differ from torch's silu * up"
        assert_fp_parity(got_np, want_np, rtol=2e-3, atol_frac=1e-5)
    # oracle: both projections unrounded-weight exact, rounded to float16 (+ bias), then the restated activation
    og, ou = exact(cg, cg["A"]), exact(cu, cu["A"])
    assert_fp_parity(g_out.cpu().numpy(), og, rtol=2e-3 if wb else 1e-3, atol_frac=6e-4)
    # the oracle's restatement of the activation on the SAME projections' outputs: numpy's exp may round the last fp32 bit unlike
    # the device's - a handful of elements one float16 ulp of silu(g) away, no more
    restated = oracle.silu_mul_f16(g_out.cpu().numpy(), u_out.cpu().numpy())
    assert int((restated.view(np.uint16) != got_np.view(np.uint16)).sum()) <= max(2, got_np.size // 200)
    assert_fp_parity(got_np, restated, rtol=2e-3, atol_frac=1e-5)
    # (fp16 outputs of the projections differ from the oracle's by their last bit here and there: compare the activation at the
    # projections' own tolerance, with the floor of an output that can cancel)
    assert_fp_parity(got_np, oracle.silu_mul_f16(og, ou).astype(np.float32), rtol=4e-3, atol_frac=2e-3)


@pytest.mark.parametrize("alias", [False, True])
@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K,wd,g,zm", SHAPES[:5])
def test_residual_add_bit_for_bit(N, K, wd, g, zm, M, alias):
    case = make_case(M, N, K, W_dtype=wd, group_size=g, with_scaling=True, with_zeros=zm is not None, zeros_mode=zm or "original",
                     with_bias=(N % 2 == 0), scale_mul=0.05, seed=N + K + M)
    mm, W, args = build(case)
    A = torch.from_numpy(case["A"]).cuda()
    rng = np.random.default_rng(N)
    res = torch.from_numpy((rng.random((M, N), dtype=np.float32) * 4 - 2).astype(np.float16)).cuda()
    # the plain result of the SAME members (the residual of zeros goes through them too), then torch's add
    plain = mm.forward_ex(A, W, residual=torch.zeros_like(res), **args)
    if M == 1:
        assert torch.equal(plain, mm(A, W, **args))        # ... which is what the operator's plain launch gives
    want = res + plain
    if alias:
        out = res.clone()
        got = mm.forward_ex(A, W, residual=out, output=out, **args)
        assert got.data_ptr() == out.data_ptr()
    else:
        got = mm.forward_ex(A, W, residual=res, **args)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert np.array_equal(got.cpu().numpy().view(np.uint16),
                          oracle.add_residual_f16(plain.cpu().numpy(), res.cpu().numpy()).view(np.uint16))
    # (with a bias the float16 result is rounded twice - cast, then the float16 bias add, as the TE graph does: an fp32 sum next to a
    # rounding boundary lands one float16 ulp = 2^-10 relative away)
    assert_fp_parity(plain.cpu().numpy(), exact(case, case["A"]), rtol=2e-3 if case["bias"] is not None else 1e-3, atol_frac=6e-4)


NORM = [  # (N, K, W_dtype, group, zeros_mode or None)
    (4096, 4096, "int4", 128, None),             # Llama-2-7B q / k / v
    (11008, 4096, "uint4", 128, "original"),     # gate / up
    (1024, 8192, "uint4", 128, "quantized"),     # Llama-3-70B hidden size: two items per thread; K split across the waves
    (2050, 2048, "int2", -1, None),
    (515, 1024, "uint1", 128, None),
]


def _hidden_and_norm(M, K, seed):
    rng = np.random.default_rng(seed)
    x = ((rng.random((M, K), dtype=np.float32) - 0.5) * 6).astype(np.float16)
    w = (1.0 + (rng.random(K, dtype=np.float32) - 0.5) * 0.5).astype(np.float16)
    return x, w


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K,wd,g,zm", NORM)
def test_rmsnorm_in_front_single_and_group(N, K, wd, g, zm, M):
    """WQAA_EPI_RMSNORM_INPUT: the operator (and a q/k/v-style group of three) fed the hidden state computes what it computes fed
    with the reference's RMSNorm output.  The fp32 sum of squares is taken in the kernel's own order: 1e-3, not bit identity."""
    case = make_case(M, N, K, W_dtype=wd, group_size=g, with_scaling=True, with_zeros=zm is not None, zeros_mode=zm or "original",
                     scale_mul=0.05, seed=N + K + M)
    mm, W, args = build(case)
    assert mm.norm_supported(M) or M * K > 12288          # (two rows of K = 8192: the selector may refuse - the entry then falls back)
    x, w = _hidden_and_norm(M, K, seed=K + M)
    xd, wd_ = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    eps = 1e-5
    normed = bitblas.matmul.rms_norm_reference(xd, wd_, eps)
    zero = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    want = mm.forward_ex(normed, W, residual=zero, **args)                      # the same family on torch's norm output
    got = mm.forward_ex(xd, W, norm=(wd_, eps), **args)
    outs = bitblas.matmul_group([mm] * 3, xd, [(W, args["scale"], args["zeros"], args["bias"])] * 3, norm=(wd_, eps))
    torch.cuda.synchronize()
    try:        # did the selector take the norm into the launch?  (two rows of K = 8192 do not fit an 8-wave workgroup's items)
        mm.lib.run_residual(xd.data_ptr(), W.data_ptr(), args["scale"].data_ptr(), args["zeros"].data_ptr() if args["zeros"] is not None else None,
                            None, torch.empty_like(got).data_ptr(), M, torch.cuda.current_stream().cuda_stream, norm=(wd_.data_ptr(), eps))
        fused = True
    except WqaaError:
        fused = False
    assert fused or K != 4096            # the Llama-2-7B shapes take it; small workgroups (few rows) or two long rows may not
    if fused:
        # the staged vector differs from torch's in a handful of float16 last bits at most (rsqrt of sums taken in different orders)
        assert_fp_parity(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-3, atol_frac=3e-4)
        for o in outs:
            assert torch.equal(o, got)                                          # a group member = the single launch, bit for bit
    else:
        assert torch.equal(got, mm(normed, W, **args))                          # torch's norm in front of the operator's plain launch
    restated = oracle.rms_norm_f16(x, w, eps)
    assert_fp_parity(normed.cpu().numpy(), restated, rtol=1e-3, atol_frac=1e-5)  # the oracle's norm = the reference's ops in torch
    assert_fp_parity(got.cpu().numpy(), exact(case, restated), rtol=1e-3, atol_frac=6e-4)


@pytest.mark.parametrize("M", [1, 2])
def test_rmsnorm_in_front_of_the_gate_up_pair(M):
    N, K = 11008, 4096
    kw = dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.08)
    cg, cu = make_case(M, N, K, seed=3 + M, **kw), make_case(M, N, K, seed=11 + M, **kw)
    gate_op, Wg, ag = build(cg)
    up_op, Wu, au = build(cu)
    assert bitblas.gate_up_plan(gate_op, M, norm=True)["name"].endswith("_pair_norm")
    x, w = _hidden_and_norm(M, K, seed=5)
    xd, wd_ = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    wg, wu = (Wg, ag["scale"], ag["zeros"], ag["bias"]), (Wu, au["scale"], au["zeros"], au["bias"])
    got = bitblas.matmul_gate_up(gate_op, up_op, xd, wg, wu, norm=(wd_, 1e-6))
    want = bitblas.matmul_gate_up(gate_op, up_op, bitblas.matmul.rms_norm_reference(xd, wd_, 1e-6), wg, wu)
    torch.cuda.synchronize()
    assert_fp_parity(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-3, atol_frac=5e-4)
    restated = oracle.rms_norm_f16(x, w, 1e-6)
    assert_fp_parity(got.cpu().numpy(), oracle.silu_mul_f16(exact(cg, restated), exact(cu, restated)).astype(np.float32), rtol=4e-3, atol_frac=2e-3)


def test_rmsnorm_where_no_fused_member_exists():
    """long K (the rows do not fit the registers a workgroup loads ahead) and formats outside the family: the Python entries run the
    reference's norm as torch kernels in front; the C entry refuses"""
    M, N, K = 1, 1024, 28672
    case = make_case(M, N, K, W_dtype="uint4", group_size=128, with_scaling=True, seed=2)
    mm, W, args = build(case)
    assert mm.fused_ops_supported(1) and not mm.norm_supported(1)
    x, w = _hidden_and_norm(M, K, seed=9)
    xd, wd_ = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    want = mm(bitblas.matmul.rms_norm_reference(xd, wd_, 1e-5), W, **args)
    assert torch.equal(mm.forward_ex(xd, W, norm=(wd_, 1e-5), **args), want)
    assert torch.equal(bitblas.matmul_group([mm, mm], xd, [(W, args["scale"])] * 2, norm=(wd_, 1e-5))[1], want)
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    with pytest.raises(WqaaError):
        mm.lib.run_residual(xd.data_ptr(), W.data_ptr(), args["scale"].data_ptr(), None, None, out.data_ptr(), 1,
                            torch.cuda.current_stream().cuda_stream, norm=(wd_.data_ptr(), 1e-5))


def test_mlp_in_two_launches():
    """x + down_proj(silu(gate_proj(h)) * up_proj(h)) of a Llama-style MLP: `matmul_gate_up` + `Linear.forward_ex` = two launches,
    against the layers' plain forwards with torch's silu, mul and add between them"""
    H, I = 4096, 11008
    rng = np.random.default_rng(3)

    def linear(n_in, n_out):
        lin = bitblas.Linear(n_in, n_out, bias=False, A_dtype="float16", W_dtype="uint4", accum_dtype="float16", out_dtype="float16",
                             group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", opt_M=[1], enable_tuning=False)
        lin.load_and_transform_weight(torch.from_numpy(rng.integers(0, 16, size=(n_out, n_in)).astype(np.int8)),
                                      scales=torch.from_numpy((rng.random((n_out, n_in // 128), dtype=np.float32) * 0.02).astype(np.float16)),
                                      zeros=torch.from_numpy(rng.integers(6, 10, size=(n_out, n_in // 128)).astype(np.float16)))
        return lin.cuda()

    gate, up, down = linear(H, I), linear(H, I), linear(I, H)
    h = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    x = torch.from_numpy((rng.random((1, H), dtype=np.float32) - 0.5).astype(np.float16)).cuda()
    act = bitblas.matmul_gate_up(gate.bitblas_matmul, up.bitblas_matmul, h, (gate.qweight, gate.scales, gate.zeros),
                                 (up.qweight, up.scales, up.zeros))
    got = down.forward_ex(act, residual=x)
    assert torch.equal(bitblas.GatedMLP(gate, up, down)(h, residual=x), got)             # the module form of the same two launches
    want_act = torch.nn.functional.silu(gate(h)) * up(h)
    want = x + down(want_act)
    torch.cuda.synchronize()
    assert (act != want_act).float().mean().item() < 0.01
    assert_fp_parity(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-3, atol_frac=5e-4)


def test_formats_without_a_fused_member():
    """nf4 has no exact-product member: the C ABI refuses loudly, the Python entries run the caller's ops as torch kernels"""
    M, N, K = 1, 1024, 1024
    case = make_case(M, N, K, W_dtype="nf4", group_size=128, with_scaling=True, seed=1)
    mm, W, args = build(case)
    assert not mm.fused_ops_supported(1) and bitblas.gate_up_plan(mm, 1) is None
    A = torch.from_numpy(case["A"]).cuda()
    res = torch.ones(M, N, dtype=torch.float16, device="cuda")
    assert torch.equal(mm.forward_ex(A, W, residual=res, **args), res + mm(A, W, **args))
    out = res.clone()
    assert torch.equal(mm.forward_ex(A, W, residual=out, output=out, **args), res + mm(A, W, **args))
    with pytest.raises(WqaaError):
        mm.lib.run_residual(A.data_ptr(), W.data_ptr(), args["scale"].data_ptr(), None, None, out.data_ptr(), 1,
                            torch.cuda.current_stream().cuda_stream, residual=res.data_ptr())
    w = (W, args["scale"])
    assert torch.equal(bitblas.matmul_gate_up(mm, mm, A, w, w), torch.nn.functional.silu(mm(A, W, **args)) * mm(A, W, **args))
    # and M = 4 of a covered format: torch kernels around the MFMA member
    case = make_case(4, N, K, W_dtype="uint4", group_size=128, with_scaling=True, seed=1)
    mm, W, args = build(case)
    assert mm.fused_ops_supported(2) and not mm.fused_ops_supported(4)
    A = torch.from_numpy(case["A"]).cuda()
    res = torch.ones(4, N, dtype=torch.float16, device="cuda")
    assert torch.equal(mm.forward_ex(A, W, residual=res, **args), res + mm(A, W, **args))
    w = (W, args["scale"])
    assert torch.equal(bitblas.matmul_gate_up(mm, mm, A, w, w), torch.nn.functional.silu(mm(A, W, **args)) * mm(A, W, **args))
