"""BitNet caller ops fused at the boundary: HIP quantiser + fused epilogue
vs the oracle restatement of integration/BitNet/utils_quant.py:150-216."""
import numpy as np
import pytest
import torch

import wqaa_oracle as oracle
from bitblas_amd import lib as wlib
from bitblas_amd.bitnet import BitLinear

pytestmark = pytest.mark.gpu


def module_codes(lin):
    """ternary weights as the module holds them (read back from the packed operand): mean|W| reduced on the GPU can
    differ from a CPU reduction in the last bit, which flips a few borderline weights - not what is under test"""
    cfg = lin.bitblas_matmul.config
    layout = wlib.LAYOUT_LOP3 if cfg.fast_decoding else wlib.LAYOUT_PLAIN
    codes = wlib.unpack_weight(lin.qweight.cpu().numpy(), lin.in_features, 2, layout, wlib.I8)
    return codes.astype(np.int8) - 2


@pytest.mark.parametrize("rows,K", [(1, 4096), (7, 1024), (300, 2560)])
def test_activation_quant_matches_reference_math(rows, K):
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, K)) * 3).astype(np.float16)
    x[0, :8] = 0
    lin = BitLinear(K, 256).cuda()
    q, s = lin.activation_quant(torch.from_numpy(x).cuda())
    wq, ws = oracle.bitnet_activation_quant(x)
    assert np.array_equal(q.cpu().numpy(), wq)
    assert np.array_equal(s.cpu().numpy(), ws[:, 0])


@pytest.mark.parametrize("m", [1, 5, 16, 64, 300])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("seed", [0, 100])
def test_bitlinear_forward(m, bias, seed):
    rng = np.random.default_rng(m + seed + 100 * bias)
    N, K = 512, 1024
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float16) if bias else None
    lin = BitLinear(K, N, bias=bias).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda(), None if b is None else torch.from_numpy(b).cuda())
    x = (rng.standard_normal((m, K))).astype(np.float16)
    got = lin(torch.from_numpy(x).cuda()).cpu().numpy()
    # sw / the ternary codes come from the module; what is under test is quantiser + matmul + fused epilogue
    wq = module_codes(lin)
    want = oracle.bitnet_forward(x, wq, np.float32(lin.sw.item()), b)
    # same integer accumulators, same two fp32 divisions, same half rounding: bit exact
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_bitlinear_llama_shape_exact():
    rng = np.random.default_rng(0)
    N, K = 4096, 4096
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    lin = BitLinear(K, N).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda())
    x = rng.standard_normal((2, K)).astype(np.float16)
    got = lin(torch.from_numpy(x).cuda()).cpu().numpy()
    wq = module_codes(lin)
    assert np.array_equal(got.view(np.uint16), oracle.bitnet_forward(x, wq, np.float32(lin.sw.item())).view(np.uint16))
    # the weight quantiser against the oracle's: same scale up to the reduction order, same codes except where
    # |w * s| sits on a rounding boundary
    wq_np, sw_np = oracle.bitnet_weight_quant(w)
    assert abs(float(sw_np) - lin.sw.item()) <= 1e-6 * float(sw_np)
    assert np.count_nonzero(wq_np != wq) <= 1e-6 * wq.size + 4


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("N,K", [(512, 1024), (4096, 4096), (1024, 11008)])
def test_single_launch_layer_for_decode_batches(m, bias, N, K):
    """m <= 4: the GEMV workgroup quantises the fp16 row itself (WQAA_EPI_QUANTIZE_INPUT): one launch for
    quantise + matmul + rescale, bit-identical to the two-launch path and to the oracle."""
    rng = np.random.default_rng(7 * m + N + K + bias)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float16) if bias else None
    lin = BitLinear(K, N, bias=bias).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda(), None if b is None else torch.from_numpy(b).cuda())
    x = (rng.standard_normal((m, K)) * 2).astype(np.float16)
    x[0, :16] = 0
    xd = torch.from_numpy(x).cuda()
    assert lin.fuse_activation_quant
    one = lin(xd).cpu().numpy()
    lin.fuse_activation_quant = False
    two = lin(xd).cpu().numpy()
    assert np.array_equal(one, two)
    wq = module_codes(lin)
    want = oracle.bitnet_forward(x, wq, np.float32(lin.sw.item()), b)
    assert np.array_equal(one, want)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_quantiser_and_layer_against_reference_run_vectors(tag):
    """tests/golden/bitnet_golden.npz holds what the reference's own activation_quant / post_quant_process
    (integration/BitNet/utils_quant.py:162-176) return on seeded inputs (oracle/gen_bitnet_golden.py runs them):
    the HIP quantiser must reproduce q and s bit for bit - including torch's `127 / t` = `reciprocal(t) * 127`."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bitnet_golden.npz"))
    x = g[f"{tag}_x"]
    K = x.shape[1]
    lin = BitLinear(K, 256).cuda()
    q, s = lin.activation_quant(torch.from_numpy(x).cuda())
    assert np.array_equal(q.cpu().numpy(), g[f"{tag}_q"])
    assert np.array_equal(s.cpu().numpy(), g[f"{tag}_si"][:, 0])


@pytest.mark.parametrize("m", [1, 2, 3, 16])
@pytest.mark.parametrize("bias", [False, True])
def test_bitlinear_group_one_launch_for_qkv(m, bias):
    """q/k/v of a BitNet block through `BitLinearGroup` (wqaa_matmul_group_ex: in-kernel quantiser + W_int2 x A_int8 +
    `out / si / sw -> half` for all three in one launch at m <= 2): bit-identical to the layers' own forward, every
    layer with its own sw, and equal to the oracle.  The reference runs three BitLinear calls or concatenates the
    weights (integration/BitNet/modeling_bitnet.py:1433-1445)."""
    from bitblas_amd.bitnet import BitLinearGroup
    rng = np.random.default_rng(100 + m + bias)
    K = 2048
    layers, raw = [], []
    for i, N in enumerate((2048, 512, 512)):
        w = (rng.standard_normal((N, K)) * 0.02 * (i + 1)).astype(np.float32)        # different mean|W| -> different sw
        b = rng.standard_normal(N).astype(np.float16) if bias else None
        lin = BitLinear(K, N, bias=bias).cuda()
        lin.load_float_weight(torch.from_numpy(w).cuda(), None if b is None else torch.from_numpy(b).cuda())
        layers.append(lin)
        raw.append(b)
    x = (rng.standard_normal((m, K)) * 2).astype(np.float16)
    xd = torch.from_numpy(x).cuda()
    want = [l(xd) for l in layers]
    got = BitLinearGroup(layers)(xd)
    torch.cuda.synchronize()
    assert len({l.sw.item() for l in layers}) == 3
    for l, b, w_, g_ in zip(layers, raw, want, got):
        assert torch.equal(w_, g_)
        ref = oracle.bitnet_forward(x, module_codes(l), np.float32(l.sw.item()), b)
        assert np.array_equal(g_.cpu().numpy(), ref)


def test_group_ex_argument_checks_and_mixed_epilogues():
    """epilogues for all members or none; members whose epilogue kinds differ run one by one with the same results"""
    import ctypes
    from bitblas_amd import group as wgroup
    rng = np.random.default_rng(5)
    K, N = 1024, 512
    lins = []
    for i in range(2):
        lin = BitLinear(K, N).cuda()
        lin.load_float_weight(torch.from_numpy((rng.standard_normal((N, K)) * 0.03).astype(np.float32)).cuda())
        lins.append(lin)
    x = torch.from_numpy((rng.standard_normal((1, K)) * 2).astype(np.float16)).cuda()
    q, si = lins[0].activation_quant(x)
    want = [l(x) for l in lins]
    L = wgroup._library()
    items = (wgroup.GroupItem * 2)()
    epis = (wlib.Epilogue * 2)()
    eptr = (ctypes.POINTER(wlib.Epilogue) * 2)()
    outs = [torch.empty_like(w) for w in want]
    for i, l in enumerate(lins):
        items[i].desc = ctypes.pointer(l.bitblas_matmul.lib.desc)
        items[i].B, items[i].C = l.qweight.data_ptr(), outs[i].data_ptr()
        epis[i].struct_size = ctypes.sizeof(wlib.Epilogue)
        epis[i].tensor_scale = float(l.sw)
        eptr[i] = ctypes.pointer(epis[i])
    # member 0: in-kernel quantiser on the float16 input; member 1: pre-quantised input + the caller's row scales
    items[0].A, epis[0].flags = x.data_ptr(), wlib.EPI_QUANTIZE_INPUT
    items[1].A, epis[1].row_scale = q.data_ptr(), si.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    assert L.wqaa_matmul_group_ex(items, eptr, 2, 1, stream) == wlib.OK
    torch.cuda.synchronize()
    for w_, o in zip(want, outs):
        assert torch.equal(w_, o)
    eptr[1] = None
    assert L.wqaa_matmul_group_ex(items, eptr, 2, 1, stream) == wlib.ERR_BAD_DESC


@pytest.mark.parametrize("m,N,K,bias", [(300, 512, 1024, True), (1000, 1024, 2048, False), (4096, 4096, 4096, True), (2048, 11008, 4096, False)])
def test_bitlinear_prefill_runs_the_ping_pong_member_with_the_epilogue(m, N, K, bias):
    """prefill row counts: the `out / si / sw -> half (+ bias)` of utils_quant.py:205-216 rides in the output stage of the
    ping-pong int2 x int8 member (round 4; it used to force the lockstep member) - the plan says so, the bits are the oracle's
    on a sample of rows"""
    rng = np.random.default_rng(m + N)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float16) if bias else None
    lin = BitLinear(K, N, bias=bias).cuda()
    lin.load_float_weight(torch.from_numpy(w).cuda(), None if b is None else torch.from_numpy(b).cuda())
    x = rng.standard_normal((m, K)).astype(np.float16)
    got = lin(torch.from_numpy(x).cuda()).cpu().numpy()
    name = lin.bitblas_matmul.lib.plan_ex(m)["name"] if hasattr(lin.bitblas_matmul.lib, "plan_ex") else ""
    rows = np.unique(np.concatenate([[0, 1, 15, 16, 127, 128, 255, 256, m - 1], rng.integers(0, m, 24)]))
    rows = rows[rows < m]
    want = oracle.bitnet_forward(x[rows], module_codes(lin), np.float32(lin.sw.item()), b)
    assert np.array_equal(got[rows].view(np.uint16), want.view(np.uint16))
    if name and m >= 2048:
        import re
        assert re.search(r"pp(t\d+)?$", name), name    # (a partial last round's columns may go out as a second launch: "ppt<n>")
