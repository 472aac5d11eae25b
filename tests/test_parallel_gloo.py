"""N > 1 path on CPU: world_size-2 gloo processes, column shard + all-gather == unsharded oracle.

The kernel launch is replaced by the oracle (no GPU here); what is under test is the sharding of
every operand form (incl. packed quantized zeros) and the gather."""
import os
import socket
import sys

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


def _worker(rank, world, port, zeros_mode, M, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wqaa_oracle as oracle
        from bitblas_amd import MatmulConfig
        from bitblas_amd.parallel import ColumnParallelMatmul, shard_operands
        rng = np.random.default_rng(7)
        N, K, g, bit = 128, 256, 64, 4
        A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
        codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
        scale = rng.random((N, K // g), dtype=np.float32).astype(np.float16)
        bias = rng.random((N,), dtype=np.float32).astype(np.float16)
        if zeros_mode == "quantized":
            zint = rng.integers(6, 10, size=(K // g, N)).astype(np.int8)
            zeros = oracle.general_compress(zint, bit)
        else:
            zeros = (8 + rng.integers(-1, 2, size=(N, K // g))).astype(np.float16)
        want = oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, zeros=zeros,
                                     zeros_mode=zeros_mode, group_size=g, bias=bias)
        cfg = MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True,
                           with_zeros=True, zeros_mode=zeros_mode, with_bias=True)
        W = torch.from_numpy(oracle.general_compress(codes, bit))     # transform_weight bytes (PLAIN)
        parts = shard_operands(rank, world, W=W, bits=bit, scale=torch.from_numpy(scale),
                               zeros=torch.from_numpy(zeros), bias=torch.from_numpy(bias), zeros_mode=zeros_mode)

        def compute(A_t, W_t, s, z, b):
            c = oracle.general_decompress(W_t.numpy(), bit)
            out = oracle.matmul_dequant(A_t.numpy(), c, source_format="uint", bit=bit, scale=s.numpy(),
                                        zeros=z.numpy(), zeros_mode=zeros_mode, group_size=g, bias=b.numpy())
            return torch.from_numpy(out)

        op = ColumnParallelMatmul(cfg, compute=compute)
        assert op.local_config.N == N // world and (op.lo, op.hi) == (rank * N // world, (rank + 1) * N // world)
        got = op(torch.from_numpy(A), parts["W"], parts["scale"], parts["zeros"], parts["bias"]).numpy()
        ret[rank] = bool(np.array_equal(got, want)) and got.shape == (M, N)
    finally:
        dist.destroy_process_group()


def _worker_dense_blocked(rank, world, port, M, row_block, ret):
    """BASELINE c5 shape family on CPU: dense e4m3 x e4m3, M rows streamed in row blocks, every block gathered +
    interleaved into the [M, N] output; the kernel launch is replaced by the oracle"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wqaa_oracle as oracle
        from bitblas_amd import MatmulConfig
        from bitblas_amd.parallel import ColumnParallelMatmul, gather_columns
        rng = np.random.default_rng(11)
        N, K = 96, 128
        A8 = torch.from_numpy(rng.random((M, K), dtype=np.float32) * 2 - 1).to(torch.float8_e4m3fn)
        W8 = torch.from_numpy(rng.random((N, K), dtype=np.float32) * 2 - 1).to(torch.float8_e4m3fn)
        want = oracle.matmul_dense(A8.view(torch.int8).numpy(), W8.view(torch.int8).numpy(), a_dtype="e4m3_float8", out_dtype="float16")
        cfg = MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")

        def compute(A_t, W_t, s, z, b):
            return torch.from_numpy(oracle.matmul_dense(A_t.view(torch.int8).numpy(), W_t.view(torch.int8).numpy(),
                                                        a_dtype="e4m3_float8", out_dtype="float16"))

        op = ColumnParallelMatmul(cfg, compute=compute, row_block=row_block)
        lo, hi = op.lo, op.hi
        got = op(A8, W8.view(torch.int8)[lo:hi].contiguous().view(torch.float8_e4m3fn))
        ok = got.shape == (M, N) and got.is_contiguous() and bool(np.array_equal(got.numpy(), want))
        # M = 1 / pre-allocated output: the collective writes straight into `out`, which is returned as is
        one = torch.from_numpy(want[:1, lo:hi].copy())
        dst = torch.empty((1, N), dtype=torch.float16)
        res = gather_columns(one, out=dst)
        ok = ok and res.data_ptr() == dst.data_ptr() and bool(np.array_equal(dst.numpy(), want[:1]))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def _worker_group(rank, world, port, M, ret):
    """q/k/v (grouped-query widths) column-sharded, one staging buffer, ONE all-gather: ColumnParallelGroup on CPU with the
    oracle in place of the kernel launch"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wqaa_oracle as oracle
        from bitblas_amd import MatmulConfig
        from bitblas_amd import parallel
        from bitblas_amd.parallel import ColumnParallelGroup, shard_operands
        rng = np.random.default_rng(5)
        K, g, bit = 256, 64, 4
        Ns = (128, 64, 32)
        A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
        cfgs, parts, wants = [], [], []
        for N in Ns:
            codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
            scale = rng.random((N, K // g), dtype=np.float32).astype(np.float16)
            wants.append(oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, group_size=g))
            cfgs.append(MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True))
            sh = shard_operands(rank, world, W=torch.from_numpy(oracle.general_compress(codes, bit)), bits=bit,
                                scale=torch.from_numpy(scale))
            parts.append((sh["W"], sh["scale"]))

        def compute(A_t, W_t, s, z, b):
            c = oracle.general_decompress(W_t.numpy(), bit)
            return torch.from_numpy(oracle.matmul_dequant(A_t.numpy(), c, source_format="uint", bit=bit, scale=s.numpy(), group_size=g))

        calls = []
        real = dist.all_gather_into_tensor
        parallel.dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            op = ColumnParallelGroup(cfgs, compute=compute)
            outs = op(torch.from_numpy(A), parts)
            views = op(torch.from_numpy(A), parts, as_views=True)
        finally:
            parallel.dist.all_gather_into_tensor = real
        ok = len(calls) == 2                                   # one collective per call, whatever the member count
        for N, o, v, w in zip(Ns, outs, views, wants):
            ok = ok and tuple(o.shape) == (M, N) and o.is_contiguous() and bool(np.array_equal(o.numpy(), w))
            ok = ok and tuple(v.shape) == (M, world, N // world) and bool(np.array_equal(v.reshape(M, N).numpy(), w))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [1, 3])
def test_column_parallel_group_one_gather_for_qkv(M):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_group, args=(world, _free_port(), M, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


@pytest.mark.parametrize("M,row_block", [(7, 512), (40, 16), (33, 8)])
def test_dense_fp8_row_blocked_gather(M, row_block):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dense_blocked, args=(r, world, port, M, row_block, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


@pytest.mark.parametrize("zeros_mode", ["original", "quantized"])
@pytest.mark.parametrize("M", [1, 5])
def test_column_shard_all_gather_matches_unsharded(zeros_mode, M):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, zeros_mode, M, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_bounds_validation():
    from bitblas_amd.parallel import shard_bounds
    assert shard_bounds(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        shard_bounds(100, 0, 8)
    with pytest.raises(ValueError):
        shard_bounds(8 * 24, 0, 8)          # 24 rows per rank: not a multiple of 16


def _worker_linear(rank, world, port, zeros_mode, ret):
    """ColumnParallelLinear: an unsharded `Linear` state_dict (the reference's checkpoint layout) loaded on two ranks, each
    keeping its rows; forward = local layer (oracle in place of the launch) + all-gather == the unsharded oracle"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wqaa_oracle as oracle
        import bitblas_amd as bitblas
        from bitblas_amd import lib as wlib
        from bitblas_amd.parallel import ColumnParallelLinear
        rng = np.random.default_rng(3)
        N, K, g, bit = 128, 256, 64, 4
        kw = dict(bias=True, A_dtype="float16", W_dtype="uint4", group_size=g, with_scaling=True, with_zeros=True,
                  zeros_mode=zeros_mode, opt_M=[1, 16], enable_tuning=False)
        full = bitblas.Linear(K, N, **kw)
        codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
        scale = rng.random((N, K // g), dtype=np.float32).astype(np.float16)
        bias = rng.random((N,), dtype=np.float32).astype(np.float16)
        if zeros_mode == "quantized":
            zeros = oracle.general_compress(rng.integers(6, 10, size=(K // g, N)).astype(np.int8), bit)
        else:
            zeros = (8 + rng.integers(-1, 2, size=(N, K // g))).astype(np.float16)
        full.load_and_transform_weight(torch.from_numpy(codes), scales=torch.from_numpy(scale), zeros=torch.from_numpy(zeros),
                                       bias=torch.from_numpy(bias))
        sd = full.state_dict()
        layer = ColumnParallelLinear(K, N, **kw)
        layer.load_full_state_dict(sd)
        lo, hi = layer.lo, layer.hi
        ok = (lo, hi) == (rank * N // world, (rank + 1) * N // world)
        ok = ok and torch.equal(layer.local.qweight, sd["qweight"][lo:hi]) and torch.equal(layer.local.scales, sd["scales"][lo:hi])
        ok = ok and torch.equal(layer.local.bias, sd["bias"][lo:hi])
        if zeros_mode == "quantized":
            ok = ok and torch.equal(layer.local.zeros, sd["zeros"][:, lo * bit // 8: hi * bit // 8])
        else:
            ok = ok and torch.equal(layer.local.zeros, sd["zeros"][lo:hi])
        layout = wlib.LAYOUT_LOP3 if layer.local.bitblas_matmul.config.fast_decoding else wlib.LAYOUT_PLAIN

        def local_forward(A_t, output=None):          # the launch, replaced by the oracle on THIS rank's buffers
            c = wlib.unpack_weight(layer.local.qweight.numpy(), K, bit, layout, wlib.F16)
            return torch.from_numpy(oracle.matmul_dequant(A_t.numpy(), c, source_format="uint", bit=bit, scale=layer.local.scales.numpy(),
                                                          zeros=layer.local.zeros.numpy(), zeros_mode=zeros_mode, group_size=g,
                                                          bias=layer.local.bias.numpy()))
        layer.local.forward = local_forward
        for M in (1, 5):
            A = (rng.random((M, K), dtype=np.float32) - 0.5).astype(np.float16)
            want = oracle.matmul_dequant(A, codes, source_format="uint", bit=bit, scale=scale, zeros=zeros, zeros_mode=zeros_mode,
                                         group_size=g, bias=bias)
            got = layer(torch.from_numpy(A))
            ok = ok and got.shape == (M, N) and bool(np.array_equal(got.numpy(), want))
        layer.gather_output = False
        ok = ok and layer(torch.from_numpy(A)).shape == (5, N // world)
        try:
            layer.shard_state_dict({"qweight": sd["qweight"][:64]})
            ok = False
        except ValueError:
            pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("zeros_mode", ["original", "quantized"])
def test_column_parallel_linear_loads_an_unsharded_checkpoint(zeros_mode):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_linear, args=(r, world, port, zeros_mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
