"""Every instantiation of the exact-product GEMV members with the callers' elementwise ops folded in (csrc/wqaa_gemvx_kernel.h,
PRO = residual add / gate-up pair / RMSNorm in front / norm + pair; include/wqaa.h: WQAA_EPI_ADD_RESIDUAL, WQAA_EPI_RMSNORM_INPUT,
wqaa_matmul_gate_up) is launched against the oracle: weight width x checkpoint layout x scale / zeros mode x batch rows x rows
per wave x fused op.  tests/test_float_ops_gpu.py checks these ops in depth on the model shapes for a few formats; a rocprofv3
census of the whole suite (tools/kernel_census.py, round 6) showed 391 of the family's 490 kernels launched by no test - the
product below launches them on small shapes (the reference's own callers: integration/BitNet/modeling_bitnet.py:240-244,
:839-860, utils_quant.py:205-216)."""
import numpy as np
import pytest
import torch

import bitblas_amd as bitblas
import wqaa_oracle as oracle
from helpers import _to_dev, assert_fp_parity, make_case

pytestmark = pytest.mark.gpu

MODES = {  # tag -> make_case keywords
    "none": dict(group_size=-1),
    "s": dict(group_size=128, with_scaling=True),
    "zo": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original"),
    "zr": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="rescale"),
    "zq": dict(group_size=128, with_scaling=True, with_zeros=True, zeros_mode="quantized"),
}
SHAPES = [(8192, 256), (384, 1024), (2048, 1024)]   # two rows per wave / one row per wave, one-wave and eight-wave workgroups (csrc/wqaa_gemvx.hip: gemvx_choose)


def _build(case):
    mm = bitblas.Matmul(case["config"], enable_tuning=False)
    W = mm.weight_transform(torch.from_numpy(case["codes"])).cuda()
    return mm, W, dict(scale=_to_dev(case["scale"], "cuda"), zeros=_to_dev(case["zeros"], "cuda"))


def _blocks(N):
    """three 128-column blocks of the output (first, middle, last): what the oracle restates - every output column depends on its
    own weight row only, and the kernels' row-group / tail logic lives at the ends"""
    return [(0, 128), ((N // 2) & ~127, ((N // 2) & ~127) + 128), (N - 128, N)] if N > 384 else [(0, N)]


def _cols(x, N):
    return np.concatenate([x[..., a:b] for a, b in _blocks(N)], axis=-1)


def _exact(case, A):
    N, bit = case["N"], case["bit"]
    outs = []
    for a, b in _blocks(N):
        zeros = case["zeros"]
        if zeros is not None:
            zeros = zeros[:, a * bit // 8:b * bit // 8] if case["zeros_mode"] == "quantized" else zeros[a:b]
        outs.append(oracle.matmul_dequant_exact(A, case["codes"][a:b], source_format=case["source_format"], bit=bit,
                                                scale=None if case["scale"] is None else case["scale"][a:b], zeros=zeros,
                                                zeros_mode=case["zeros_mode"], group_size=case["g"], out_dtype="float16"))
    return np.concatenate(outs, axis=-1)


def _case(M, N, K, bits, fd, mode, seed):
    kw = dict(MODES[mode])
    # (unscaled codes: keep the sums inside float16 - the operands of the reference's own tests are rand - 0.5 as well)
    return make_case(M, N, K, W_dtype=f"uint{bits}", fast_decoding=fd, scale_mul=0.05, seed=seed, **kw)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K", SHAPES)
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("fd", [None, False], ids=["lop3", "plain"])
@pytest.mark.parametrize("bits", [4, 2, 1])
def test_residual_norm_and_pair_members(bits, fd, mode, N, K, M):
    seed = bits * 1000 + N + K + M
    cg, cu = _case(M, N, K, bits, fd, mode, seed), _case(M, N, K, bits, fd, mode, seed + 7)
    cu["A"] = cg["A"]
    gate, Wg, ag = _build(cg)
    up, Wu, au = _build(cu)
    if not gate.fused_ops_supported(M):
        pytest.skip("the selector keeps this shape on a family without fused members")
    A = torch.from_numpy(cg["A"]).cuda()
    rng = np.random.default_rng(seed)
    # ---- residual add: bit for bit the plain result + torch's add; the plain result against the oracle ----
    res = torch.from_numpy((rng.random((M, N), dtype=np.float32) * 4 - 2).astype(np.float16)).cuda()
    plain = gate.forward_ex(A, Wg, residual=torch.zeros_like(res), **ag)
    got = gate.forward_ex(A, Wg, residual=res, **ag)
    torch.cuda.synchronize()
    assert torch.equal(got, res + plain)
    og = _exact(cg, cg["A"])
    assert_fp_parity(_cols(plain.cpu().numpy(), N), og, rtol=1e-3, atol_frac=6e-4)
    # ---- gate / up pair ----
    wg, wu = (Wg, ag["scale"], ag["zeros"], None), (Wu, au["scale"], au["zeros"], None)
    act = bitblas.matmul_gate_up(gate, up, A, wg, wu)
    torch.cuda.synchronize()
    ou = _exact(cu, cu["A"])
    assert_fp_parity(_cols(act.cpu().numpy(), N), oracle.silu_mul_f16(og, ou).astype(np.float32), rtol=4e-3, atol_frac=2e-3)
    # ---- RMSNorm in front, alone and in front of the pair ----
    if not gate.norm_supported(M):
        return
    x = ((rng.random((M, K), dtype=np.float32) - 0.5) * 6).astype(np.float16)
    # (unscaled codes: a small norm weight keeps silu(gate) * up inside float16)
    w = ((1.0 + (rng.random(K, dtype=np.float32) - 0.5) * 0.5) * (0.05 if mode == "none" else 1.0)).astype(np.float16)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    eps = 1e-5
    restated = oracle.rms_norm_f16(x, w, eps)
    normed = gate.forward_ex(xd, Wg, norm=(wd, eps), **ag)
    torch.cuda.synchronize()
    # (the kernel's rsqrt of a sum taken in its own order moves a few staged activations by a float16 ulp: at K = 512 / 1024 that
    # shows as 1-2 ulp on a handful of outputs - tests/test_float_ops_gpu.py holds the model shapes to 1e-3 + 6e-4)
    assert_fp_parity(_cols(normed.cpu().numpy(), N), _exact(cg, restated), rtol=2e-3, atol_frac=2e-3)
    pair = bitblas.matmul_gate_up(gate, up, xd, wg, wu, norm=(wd, eps))
    torch.cuda.synchronize()
    assert_fp_parity(_cols(pair.cpu().numpy(), N), oracle.silu_mul_f16(_exact(cg, restated), _exact(cu, restated)).astype(np.float32), rtol=6e-3, atol_frac=4e-3)
