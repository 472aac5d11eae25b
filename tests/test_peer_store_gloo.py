"""The M = 1 peer-store path of ColumnParallelMatmul (direct_store=True) with world_size-2 gloo processes on the CPU.

What is under test is the bookkeeping every transport shares (bitblas_amd/peer.py PeerWindow: steps, row slots, the slice
offsets a rank stores to and the flag words it posts / waits on) - through the shared-mapping transport, with the oracle in
place of the kernel launch: no collective on the path, every rank ends up with the unsharded oracle's row, a returned row
stays intact until the next call is made, however far ahead the peer is (two slots), a rank that stops posting is reported instead of waited for forever.  The hipIpc transport
(csrc/wqaa_peer.hip) runs the same bookkeeping on a GPU box: tests/test_parallel_gpu.py."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_direct(rank, world, port, ret):
    _setup(rank, world, port)
    try:
        import wqaa_oracle as oracle
        from bitblas_amd import MatmulConfig
        from bitblas_amd import parallel
        from bitblas_amd.parallel import ColumnParallelMatmul, shard_operands
        rng = np.random.default_rng(3)
        N, K, g, bit = 256, 256, 64, 4
        codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
        scale = rng.random((N, K // g), dtype=np.float32).astype(np.float16)
        cfg = MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True)
        sh = shard_operands(rank, world, W=torch.from_numpy(oracle.general_compress(codes, bit)), bits=bit, scale=torch.from_numpy(scale))

        def compute(A_t, W_t, s, z, b):
            c = oracle.general_decompress(W_t.numpy(), bit)
            return torch.from_numpy(oracle.matmul_dequant(A_t.numpy(), c, source_format="uint", bit=bit, scale=s.numpy(), group_size=g))

        calls = []
        real = dist.all_gather_into_tensor
        parallel.dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            op = ColumnParallelMatmul(cfg, compute=compute, direct_store=True, row_block=512)
            ok = True
            kept = []
            for step in range(1, 8):
                A = (rng.random((1, K), dtype=np.float32) - 0.5).astype(np.float16)     # same seed on both ranks: same input
                want = oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, group_size=g)
                if rank == 1 and step == 3:
                    time.sleep(0.2)                                                   # a slow rank: its peer has to wait, not read early
                if kept:                                                              # the previous call's row is intact until the next call is made:
                    if rank == 0 and step == 5:                                       # a fast peer already works on the next step - in the other slot
                        time.sleep(0.2)
                    ok = ok and bool(np.array_equal(kept[-1][0].numpy(), kept[-1][1]))
                got = op(torch.from_numpy(A), sh["W"], sh["scale"])
                ok = ok and tuple(got.shape) == (1, N) and bool(np.array_equal(got.numpy(), want))
                ok = ok and op._window.step == step and op._window.slot_of(step) == (step - 1) % 2
                kept.append((got, want))
                if step == 1:
                    calls.clear()                                                     # (the window's set-up exchanged its handles)
            # out=: a copy, detached from the window
            A = (rng.random((1, K), dtype=np.float32) - 0.5).astype(np.float16)
            dst = torch.empty((1, N), dtype=torch.float16)
            res = op(torch.from_numpy(A), sh["W"], sh["scale"], out=dst)
            ok = ok and res.data_ptr() == dst.data_ptr()
            # more than one row: the collective path, unchanged
            A3 = (rng.random((3, K), dtype=np.float32) - 0.5).astype(np.float16)
            got3 = op(torch.from_numpy(A3), sh["W"], sh["scale"])
            want3 = oracle.matmul_dequant(A3, codes, source_format="uint", bit=bit, scale=scale, group_size=g)
            ok = ok and bool(np.array_equal(got3.numpy(), want3))
            op.check_peers()
        finally:
            parallel.dist.all_gather_into_tensor = real
        ret[rank] = (ok and len(calls) == 1) or f"ok={ok} gathers={len(calls)}"     # only the three-row call gathered
    finally:
        dist.destroy_process_group()


def _worker_window(rank, world, port, ret):
    """the window on its own: offsets, slice checks, wrap-around of the step compare, a peer that never posts"""
    _setup(rank, world, port)
    try:
        from bitblas_amd.peer import FLAG_REGION, PeerTimeout, ShmPeerWindow
        win = ShmPeerWindow(None, row_bytes=96, slots=3, timeout_ms=1500)   # (generous: a loaded host must not make the live peer look silent)
        ok = win.row_pitch == 256 and win.window_bytes == FLAG_REGION + 3 * 256 and win.row_offset(2) == FLAG_REGION + 512
        for bad in ((8, 16), (0, 24), (96, 16), (-16, 16), (0, 0)):
            try:
                win.exchange(1, *bad)
                ok = False
            except ValueError:
                pass
        # steps near the 32-bit wrap: both ranks jump there together
        win.step = 0xFFFFFFFE - 1
        for _ in range(4):
            s = win.next_step()
            row = win.row(win.slot_of(s), torch.uint8)
            row[rank * 48:(rank + 1) * 48] = (s + rank) & 0xFF
            win.exchange(s, rank * 48, 48)
            other = 1 - rank
            ok = ok and bool((row[other * 48:(other + 1) * 48] == ((s + other) & 0xFF)).all())
        win.check()
        dist.barrier()
        # rank 1 stops: rank 0's next exchange gives up after the timeout and says who was late
        if rank == 0:
            s = win.next_step()
            t0 = time.monotonic()
            win.exchange(s, 0, 48)
            waited = time.monotonic() - t0
            try:
                win.check()
                ok = False
            except PeerTimeout as e:
                ok = ok and "rank 1" in str(e) and 1.2 < waited < 30.0
        dist.barrier()
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def _run(worker):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) is True for r in range(world)), dict(ret)


def test_single_row_is_exchanged_through_peer_windows_without_a_collective():
    _run(_worker_direct)


def test_window_layout_wraparound_and_a_silent_peer():
    _run(_worker_window)


def test_window_argument_checks():
    from bitblas_amd.peer import PeerWindow

    class G:
        pass
    import torch.distributed as d
    if d.is_initialized():
        pytest.skip("needs no process group")
    # (constructor checks need a group only for world / rank: exercise the arithmetic through a stub)
    w = PeerWindow.__new__(PeerWindow)
    w.row_bytes, w.slots, w.row_pitch = 512, 2, 512
    assert [w.slot_of(s) for s in (1, 2, 3, 4)] == [0, 1, 0, 1]
    with pytest.raises(IndexError):
        w.row_offset(2)
    with pytest.raises(ValueError):
        w._check_slice(0, 520)
