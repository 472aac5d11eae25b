"""The N > 1 device path on ONE GPU: two processes share cuda:0 (gloo carries the CUDA tensors - RCCL refuses two ranks
on one device), so the row-blocked pipeline of ColumnParallelMatmul - HIP GEMM of block i + 1 on the compute stream,
all-gather + interleave of block i on the communication stream, staging buffers reused - runs with real kernels and is
compared with the oracle.  (World-size-2 CPU tests of the sharding itself: tests/test_parallel_gloo.py.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wqaa_oracle as oracle
        from bitblas_amd import MatmulConfig
        from bitblas_amd.parallel import ColumnParallelMatmul
        torch.cuda.set_device(0)
        probe = torch.zeros(4, device="cuda")
        try:
            dist.all_gather_into_tensor(torch.zeros(4 * world, device="cuda"), probe)
        except Exception as e:  # noqa: BLE001
            ret[rank] = f"skip: gloo cannot gather CUDA tensors here ({type(e).__name__})"
            return
        rng = np.random.default_rng(5)
        M, N, K = 1200, 1024, 1024
        A8 = torch.from_numpy(rng.random((M, K), dtype=np.float32) * 2 - 1).to(torch.float8_e4m3fn)
        W8 = torch.from_numpy(rng.random((N, K), dtype=np.float32) * 2 - 1).to(torch.float8_e4m3fn)
        cfg = MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")
        op = ColumnParallelMatmul(cfg, row_block=512)                 # blocks of 512, 512, 176 rows
        Wl = W8.view(torch.int8)[op.lo:op.hi].contiguous().view(torch.float8_e4m3fn).cuda()
        out = torch.empty((M, N), dtype=torch.float16, device="cuda")
        for _ in range(3):                                            # staging reuse across calls
            got = op(A8.cuda(), Wl, out=out)
        torch.cuda.synchronize()
        rows = np.arange(0, M, 7)
        want = oracle.matmul_dense(A8.view(torch.int8).numpy()[rows], W8.view(torch.int8).numpy(), a_dtype="e4m3_float8",
                                   out_dtype="float32")
        g = got.float().cpu().numpy()[rows]
        err = np.abs(g - want)
        ok = got.data_ptr() == out.data_ptr() and bool((err <= 1e-3 * np.abs(want) + 1e-3 * np.sqrt(np.mean(want ** 2))).all())
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_row_blocked_gather_overlaps_with_real_kernels():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    vals = [ret.get(r) for r in range(world)]
    if any(isinstance(v, str) and v.startswith("skip") for v in vals):
        pytest.skip(str(vals))
    assert all(v is True for v in vals), vals


def _worker_group(rank, world, port, ret):
    """ColumnParallelGroup with the real kernels: q/k/v (grouped-query widths) of an int4 layer, two ranks on cuda:0, one
    group launch per rank + one all-gather, compared with the oracle and with the members run unsharded"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bitblas_amd as bitblas
        import wqaa_oracle as oracle
        from bitblas_amd.parallel import ColumnParallelGroup, shard_operands
        torch.cuda.set_device(0)
        probe = torch.zeros(4, device="cuda")
        try:
            dist.all_gather_into_tensor(torch.zeros(4 * world, device="cuda"), probe)
        except Exception as e:  # noqa: BLE001
            ret[rank] = f"skip: gloo cannot gather CUDA tensors here ({type(e).__name__})"
            return
        rng = np.random.default_rng(9)
        K, g, bit, M = 2048, 128, 4, 1
        Ns = (2048, 512, 512)
        A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
        cfgs, parts, wants = [], [], []
        for N in Ns:
            codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
            scale = (rng.random((N, K // g), dtype=np.float32) * 0.05).astype(np.float16)
            wants.append(oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, group_size=g))
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True)
            cfgs.append(cfg)
            full = bitblas.Matmul(cfg, enable_tuning=False)
            W = full.weight_transform(torch.from_numpy(codes))               # the reference's checkpoint bytes, then sharded
            sh = shard_operands(rank, world, W=W, bits=bit, scale=torch.from_numpy(scale))
            parts.append((sh["W"].cuda(), sh["scale"].cuda()))
        op = ColumnParallelGroup(cfgs)
        from bitblas_amd import group_plan
        fused = group_plan(op.ops, M)["launches"] == 1
        outs = op(torch.from_numpy(A).cuda(), parts)
        torch.cuda.synchronize()
        ok = fused
        for N, o, w in zip(Ns, outs, wants):
            got = o.float().cpu().numpy()
            err = np.abs(got - w)
            ok = ok and tuple(o.shape) == (M, N) and bool((err <= 1e-3 * np.abs(w) + 1.5e-3 * np.sqrt(np.mean(w.astype(np.float64) ** 2))).all())
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_column_parallel_group_with_real_kernels():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_group, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    vals = [ret.get(r) for r in range(world)]
    if any(isinstance(v, str) and v.startswith("skip") for v in vals):
        pytest.skip(str(vals))
    assert all(v is True for v in vals), vals


def _worker_direct_store(rank, world, port, ret):
    """ColumnParallelMatmul(direct_store=True) with the real kernels and the hipIpc transport: two ranks on cuda:0 map each
    other's window (csrc/wqaa_peer.hip), the M = 1 GEMV writes its slice into the own window row, ONE exchange launch per
    step stores it into the peer's row and waits for the peer's - no collective after the set-up"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bitblas_amd as bitblas
        import wqaa_oracle as oracle
        from bitblas_amd import parallel
        from bitblas_amd.lib import WqaaError
        from bitblas_amd.parallel import ColumnParallelMatmul, shard_operands
        torch.cuda.set_device(0)
        rng = np.random.default_rng(13)
        N, K, g, bit = 4096, 4096, 128, 4
        codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
        scale = (rng.random((N, K // g), dtype=np.float32) * 0.05).astype(np.float16)
        cfg = bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True)
        full = bitblas.Matmul(cfg, enable_tuning=False)
        sh = shard_operands(rank, world, W=full.weight_transform(torch.from_numpy(codes)), bits=bit, scale=torch.from_numpy(scale))
        Wl, Sl = sh["W"].cuda(), sh["scale"].cuda()
        op = ColumnParallelMatmul(cfg, direct_store=True)
        calls = []
        real = dist.all_gather_into_tensor
        ok = True
        try:
            for step in range(1, 7):
                A = (rng.random((1, K), dtype=np.float32) - 0.5).astype(np.float16)
                want = oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, group_size=g)
                try:
                    got = op(torch.from_numpy(A).cuda(), Wl, Sl)
                except WqaaError as e:
                    ret[rank] = f"skip: no IPC mapping between two processes here ({e})"
                    return
                if step == 1:
                    parallel.dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
                g_ = got.float().cpu().numpy()                     # (stream-ordered read of the window row: before the next call)
                err = np.abs(g_ - want)
                ok = ok and tuple(got.shape) == (1, N)
                ok = ok and bool((err <= 1e-3 * np.abs(want) + 1.5e-3 * np.sqrt(np.mean(want.astype(np.float64) ** 2))).all())
            op.check_peers()
        finally:
            parallel.dist.all_gather_into_tensor = real
        ret[rank] = (ok and not calls) or f"ok={ok} gathers={len(calls)}"
        dist.barrier()
        op._window.close()
    finally:
        dist.destroy_process_group()


def test_single_row_direct_store_through_ipc_windows():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_direct_store, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    vals = [ret.get(r) for r in range(world)]
    if any(isinstance(v, str) and v.startswith("skip") for v in vals):
        pytest.skip(str(vals))
    assert all(v is True for v in vals), vals
