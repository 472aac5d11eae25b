"""Groups of operators in one launch (wqaa_matmul_group): what can be checked without a GPU.

* the GEMV kernels' workgroup -> row-group-block map (csrc/wqaa_kinds.h `xcd_row_blocks`, through its host twin
  `wqaa_debug_row_blocks`): every block is worked on exactly once for any grid, and the eight XCDs get equal shares;
* which groups the library fuses (`wqaa_group_plan`): the q/k/v and gate/up projections of a decoder layer - the
  reference fuses the same projections by concatenating their weights, integration/BitNet/modeling_bitnet.py:1433-1445 -
  and which it runs member by member;
* argument checking of the C entry.
"""
import ctypes

import pytest

import bitblas_amd as bitblas
from bitblas_amd import group as wgroup
from bitblas_amd import lib as wlib


def row_blocks(b, grid, n_blocks):
    L = wlib.load_library()
    out = (ctypes.c_int * 3)()
    L.wqaa_debug_row_blocks(int(b), int(grid), int(n_blocks), out)
    first, stride, end = out[0], out[1], out[2]
    return list(range(first, end, stride)) if first < end else []


@pytest.mark.parametrize("grid,n_blocks", [
    (512, 512), (688, 688), (696, 688),            # one block per workgroup (the usual launch), with grid padding
    (1024, 1376), (2048, 3584), (8, 1000), (16, 17),   # fewer workgroups than blocks: several rounds
    (512, 128), (768, 64), (64, 1), (8, 3),        # a short member of a group launch: more workgroups than blocks
    (5, 23), (7, 7), (3, 1), (1, 9),               # grids that are no multiple of 8: plain striding
])
def test_row_block_map_covers_every_block_once(grid, n_blocks):
    seen = []
    per_xcd = [0] * 8
    for b in range(grid):
        mine = row_blocks(b, grid, n_blocks)
        seen += mine
        per_xcd[b % 8] += len(mine)
    assert sorted(seen) == list(range(n_blocks))
    if grid % 8 == 0:
        # XCD x owns the contiguous eighth [x * ceil(n / 8), ...): equal shares up to the rounding of the last one
        chunk = (n_blocks + 7) // 8
        assert max(per_xcd) <= chunk
        assert sum(1 for c in per_xcd if c == chunk) >= min(8, n_blocks // chunk)
        for b in range(grid):
            for blk in row_blocks(b, grid, n_blocks):
                assert blk // chunk == b % 8


def test_row_block_map_is_the_plain_swizzle_when_grid_matches():
    # grid = blocks rounded up to 8: blk = (b & 7) * (grid >> 3) + (b >> 3), one block per workgroup
    for n_blocks in (512, 688, 1371):
        grid = (n_blocks + 7) // 8 * 8
        for b in range(grid):
            want = (b & 7) * (grid >> 3) + (b >> 3)
            assert row_blocks(b, grid, n_blocks) == ([want] if want < n_blocks else [])


def op(N, K=4096, M=1, W_dtype="int4", strict=False, **kw):
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype=W_dtype, accum_dtype="float16",
                               out_dtype="float16", group_size=128, with_scaling=True, **kw)
    return bitblas.Matmul(cfg, enable_tuning=False, strict_reference=strict)


@pytest.mark.parametrize("strict", [False, True])
def test_decoder_layer_projections_fuse(strict):
    family = "_gemv_" if strict else "_gemvx_"
    qkv = wgroup.group_plan([op(4096, strict=strict)] * 3, 1)
    assert qkv["launches"] == 1
    # the tile configuration is the merged operator's (what a concatenated qkv Linear would get)
    merged = op(12288, strict=strict).plans[1]
    assert qkv["plan"]["name"] == merged["name"] + "_x3" and family in merged["name"]
    assert qkv["plan"]["threads"] == merged["threads"] and qkv["plan"]["lds_bytes"] == merged["lds_bytes"]
    gate_up = wgroup.group_plan([op(11008, strict=strict)] * 2, 1)
    assert gate_up["launches"] == 1 and gate_up["plan"]["name"].endswith("_x2")
    assert gate_up["plan"]["grid"] % 16 == 0          # every member: whole XCD rounds
    # grouped-query attention: k / v narrower than q
    gqa = wgroup.group_plan([op(4096, strict=strict), op(1024, strict=strict), op(1024, strict=strict)], 1)
    assert gqa["launches"] == 1 and "n6144" in gqa["plan"]["name"]
    assert wgroup.group_plan([op(4096, strict=strict)] * 2, 2)["launches"] == 1


def test_groups_that_run_member_by_member():
    a, b = op(4096), op(4096, K=8192)
    assert wgroup.group_plan([a, b], 1) == {"launches": 2, "plan": None}            # different K
    assert wgroup.group_plan([op(4096), op(4096, W_dtype="uint4")], 1)["launches"] == 2   # different format
    assert wgroup.group_plan([op(4096, strict=True), op(4096, strict=False)], 1)["launches"] == 2
    dyn = [op(4096, M=[1, 16]), op(4096, M=[1, 16])]
    assert wgroup.group_plan(dyn, 1)["launches"] == 1
    assert wgroup.group_plan(dyn, 16)["launches"] == 2                               # MFMA members are not fused
    assert wgroup.group_plan([op(4096)], 1)["launches"] == 1                         # a group of one: the plain call
    assert wgroup.group_plan([op(256)] * 9, 1)["launches"] == 9                      # beyond WQAA_GROUP_MAX


def test_group_entry_argument_checks():
    L = wgroup._library()
    assert ctypes.sizeof(wgroup.GroupItem) == 8 * ctypes.sizeof(ctypes.c_void_p)
    assert L.wqaa_matmul_group(None, 2, 1, None) == wlib.ERR_BAD_DESC
    items = (wgroup.GroupItem * 2)()
    assert L.wqaa_matmul_group(items, 0, 1, None) == wlib.OK            # empty group
    assert L.wqaa_matmul_group(items, 2, 0, None) == wlib.OK            # m == 0 (wrapper/tl.py:277)
    assert L.wqaa_matmul_group(items, 2, 1, None) == wlib.ERR_BAD_DESC  # null descriptors
    d = op(4096).lib.desc
    for it in items:
        it.desc = ctypes.pointer(d)
    assert L.wqaa_matmul_group(items, 2, 1, None) == wlib.ERR_BAD_DESC  # null operands, refused before any launch
    assert b"member 0" in L.wqaa_last_error_string()


def test_matmul_group_validates_before_it_launches(monkeypatch):
    """`matmul_group` on CPU tensors with the two device-touching calls replaced: a valid call reaches the C entry with every
    member's own pointers; wrong row counts, short or mistyped outputs and short weights are refused before it"""
    import torch
    ops = [bitblas.Matmul(bitblas.MatmulConfig(M=1, N=n, K=256, A_dtype="float16", W_dtype="int4", group_size=128, with_scaling=True),
                          enable_tuning=False, strict_reference=False) for n in (128, 64)]
    for op in ops:
        monkeypatch.setattr(op, "check_activation", lambda A: A.numel() // A.shape[-1], raising=False)
    monkeypatch.setattr(wgroup._lib, "current_stream_handle", lambda dev: 0)
    seen = []

    class FakeLib:
        def wqaa_matmul_group(self, items, n, m, stream):
            seen.append([(items[i].A, items[i].B, items[i].Scale, items[i].C, items[i].desc.contents.N) for i in range(n)] + [m])
            return wlib.OK
    monkeypatch.setattr(wgroup, "_library", lambda: FakeLib())
    A = torch.zeros(1, 256, dtype=torch.float16)
    Ws = [(torch.zeros(n, 128, dtype=torch.int8), torch.zeros(n, 2, dtype=torch.float16)) for n in (128, 64)]
    outs = bitblas.matmul_group(ops, A, Ws)
    assert [tuple(o.shape) for o in outs] == [(1, 128), (1, 64)] and len(seen) == 1 and seen[0][-1] == 1
    for (a, b, s, c, n), (W, sc), o in zip(seen[0][:-1], Ws, outs):
        assert (a, b, s, c) == (A.data_ptr(), W.data_ptr(), sc.data_ptr(), o.data_ptr()) and n == o.shape[1]
    mine = [torch.empty(1, 128, dtype=torch.float16), torch.empty(64, dtype=torch.float16)]      # any contiguous view of m x N
    assert bitblas.matmul_group(ops, A, Ws, outputs=mine)[1] is mine[1]
    with pytest.raises(ValueError, match="output must hold"):
        bitblas.matmul_group(ops, A, Ws, outputs=[torch.empty(1, 128, dtype=torch.float16), torch.empty(1, 32, dtype=torch.float16)])
    with pytest.raises(ValueError, match="output must hold"):
        bitblas.matmul_group(ops, A, Ws, outputs=[torch.empty(1, 128, dtype=torch.float32), torch.empty(1, 64, dtype=torch.float16)])
    with pytest.raises(ValueError, match="W holds"):
        bitblas.matmul_group(ops, A, [(Ws[0][0][:64], Ws[0][1]), Ws[1]])
    with pytest.raises(ValueError, match="weights"):
        bitblas.matmul_group(ops, A, Ws[:1])
    assert len(seen) == 2                                                                     # none of the refused calls launched


def test_gate_up_pair_plan_and_argument_checks():
    """wqaa_matmul_gate_up / wqaa_gate_up_plan (include/wqaa.h) without a device: the pair launch takes the tile configuration of the
    two projections concatenated, two rows per wave; what the exact-product family does not cover is refused"""
    gate = op(11008)
    plan = wgroup.gate_up_plan(gate, 1)
    merged = op(22016).plans[1]
    assert plan["name"] == gate.plans[1]["name"].replace("_gemvx_b1r2d2k1", "_gemvx_b1r2d2k1_pair") and plan["rows_per_wave"] == 2
    assert plan["threads"] == merged["threads"] and plan["split_k"] == merged["split_k"] and plan["lds_bytes"] == merged["lds_bytes"]
    assert wgroup.gate_up_plan(op(512, K=16384), 1)["split_k"] > 1            # few rows: K split across the waves, like the plain launch
    for n, k in ((512, 16384), (1000, 16384), (300, 8192), (2048, 12288), (11008, 4096)):      # a row is summed as its projection alone sums it
        alone = op(n, K=k)
        assert wgroup.gate_up_plan(alone, 1)["split_k"] == alone.plans[1]["split_k"], (n, k)
    assert wgroup.gate_up_plan(op(11008, M=[1, 16]), 2)["batch_tile"] == 2
    assert wgroup.gate_up_plan(op(11008, M=[1, 16]), 16) is None              # MFMA row counts: the caller's own elementwise kernels
    assert wgroup.gate_up_plan(op(4096, W_dtype="nf4"), 1) is None
    L = wgroup._library()
    items = (wgroup.GroupItem * 2)()
    assert L.wqaa_matmul_gate_up(None, None, None, 1, None, None) == wlib.ERR_BAD_DESC
    d = gate.lib.desc
    for it in items:
        it.desc = ctypes.pointer(d)
    assert L.wqaa_matmul_gate_up(ctypes.byref(items[0]), ctypes.byref(items[1]), None, 0, None, None) == wlib.OK        # m == 0
    assert L.wqaa_matmul_gate_up(ctypes.byref(items[0]), ctypes.byref(items[1]), None, 1, None, None) == wlib.ERR_BAD_DESC
    other = op(4096).lib.desc
    items[1].desc = ctypes.pointer(other)
    assert L.wqaa_matmul_gate_up(ctypes.byref(items[0]), ctypes.byref(items[1]), None, 1, None, None) == wlib.ERR_BAD_DESC
    assert b"agree" in L.wqaa_last_error_string()


def test_epilogue_descriptor_layout():
    """struct wqaa_epilogue grew by the residual and norm fields; its 24-byte prefix is what callers built before that pass"""
    assert ctypes.sizeof(wlib.Epilogue) == 48 and wlib.Epilogue.residual.offset == 24 and wlib.Epilogue.norm_weight.offset == 32
    assert wlib.Epilogue.norm_eps.offset == 40
    assert wlib.EPI_QUANTIZE_INPUT == 1 and wlib.EPI_ADD_RESIDUAL == 2 and wlib.EPI_RMSNORM_INPUT == 4
    # the norm in front: plans without a device; K beyond the registers a workgroup loads ahead is refused
    assert wgroup.gate_up_plan(op(11008), 1, norm=True)["name"].endswith("_pair_norm")
    assert wgroup.gate_up_plan(op(4096, K=28672), 1, norm=True) is None and wgroup.gate_up_plan(op(4096, K=28672), 1) is not None
    two_rows = op(4096, K=8192, M=[1, 2])      # two rows of K = 8192 against an 8-wave workgroup's two items per thread
    assert wgroup.gate_up_plan(two_rows, 1, norm=True) is not None and wgroup.gate_up_plan(two_rows, 2, norm=True) is None
    assert op(4096).norm_supported(1) and not op(4096, K=28672).norm_supported(1) and not op(4096, W_dtype="nf4").norm_supported(1)
    mm = op(4096)
    assert mm.fused_ops_supported(1) and mm.fused_ops_supported(2) and not mm.fused_ops_supported(3)
    assert not op(4096, W_dtype="nf4").fused_ops_supported(1)


@pytest.mark.parametrize("tiles_m,tiles_n,ksplit,group_m", [(16, 16, 1, 4), (1, 32, 8, 1), (2, 32, 4, 1), (5, 43, 1, 4), (7, 3, 2, 4), (9, 86, 1, 4),
                                                            (32, 224, 1, 4), (1, 1, 1, 1), (3, 7, 16, 1), (13, 5, 3, 4), (6, 11, 1, 8)])
def test_mfma_tile_map_without_divisions(tiles_m, tiles_n, ksplit, group_m):
    """the MFMA members' workgroup -> (k-slice, M-tile, N-tile) map (csrc/wqaa_gemm_kernel.h: tile_of_block, host twin
    `wqaa_debug_tile_of_block`) runs on host-side reciprocals instead of integer divisions: every tile of every k-slice is taken
    exactly once, and the answer is the one the division-based definition gives"""
    L = wlib.load_library()
    L.wqaa_debug_tile_of_block.restype = None
    L.wqaa_debug_tile_of_block.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)]
    out = (ctypes.c_int * 4)()
    nblocks = tiles_m * tiles_n * ksplit
    seen = set()
    for b in range(nblocks):
        L.wqaa_debug_tile_of_block(tiles_m, tiles_n, ksplit, group_m, b, out)
        blk = (b & 7) * (nblocks >> 3) + (b >> 3) if nblocks % 8 == 0 else b
        split, rest = divmod(blk, tiles_m * tiles_n)
        grp, rem = divmod(rest, group_m * tiles_n)
        first_m = grp * group_m
        gsz = min(group_m, tiles_m - first_m)
        assert (out[0], out[1], out[2]) == (split, first_m + rem % gsz, rem // gsz), (b, list(out))
        assert out[3] == 1 or tiles_m * tiles_n == 1
        seen.add((out[0], out[1], out[2]))
    assert seen == {(s, m, n) for s in range(ksplit) for m in range(tiles_m) for n in range(tiles_n)}
