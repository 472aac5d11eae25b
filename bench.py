#!/usr/bin/env python
"""bench.py - the headline measurement (BASELINE.json): W_int4 A_fp16 GEMV at M=1 on the
Llama-2-7B linear shapes, group_size=128, on MI355X.  One JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one decode pass over LAYERS synthetic decoder layers: each layer = the 7 linear GEMVs of a
Llama-2-7B block (q,k,v,o: 4096x4096; gate,up: 11008x4096; down: 4096x11008), every layer with its
own weight buffers, so a step streams LAYERS x 105 MB of distinct packed weights (> the 256 MiB
Infinity Cache): the bytes really come from HBM.  Inputs are resident in HBM before the timed
region.  value = algorithmic bytes moved per second by the whole job (GB/s).

roofline: the dominant kernel (the int4 GEMV) is timed per launch from the kernel's own begin/end
timestamps (hipExtLaunchKernel events through `wqaa_matmul_timed`), on the 4096x4096 member, again
rotating over enough buffers to defeat the Infinity Cache; `achieved` = algorithmic bytes / mean
duration.  Also reported: the MFMA GEMM member at M=4096 (TFLOP/s) when the library has one.

N > 1: the weight matrices are column(N)-sharded: every rank owns an equal slice of every layer
(weak scaling: per-GPU work fixed, i.e. the model is N_gpus x wider) and the per-layer output slices
are all-gathered with one RCCL all-gather per step on a side stream.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bitblas_amd as bitblas  # noqa: E402
from bitblas_amd import lib as wlib  # noqa: E402

LLAMA2_7B_LINEARS = [  # (name, N, K)
    ("q_proj", 4096, 4096), ("k_proj", 4096, 4096), ("v_proj", 4096, 4096), ("o_proj", 4096, 4096),
    ("gate_proj", 11008, 4096), ("up_proj", 11008, 4096), ("down_proj", 4096, 11008),
]
GROUP = 128
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0   # dense fp16/bf16 MFMA peak


def algorithmic_bytes(M, N, K, bits=4, g=GROUP, scale=True, zeros=False, out_bytes=2, a_bytes=2):
    """M*K*sA + N*K*bit/8 + N*(K/g)*2 [scale] + N*(K/g)*2 [zeros] + M*N*sOut"""
    b = M * K * a_bytes + N * K * bits // 8 + M * N * out_bytes
    if scale:
        b += N * (K // g) * 2
    if zeros:
        b += N * (K // g) * 2
    return b


class Hip:
    """The few HIP runtime calls the bench needs (events with kernel-level timestamps)."""

    def __init__(self):
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.rt.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.rt.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def event(self):
        e = ctypes.c_void_p()
        assert self.rt.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        assert self.rt.hipEventSynchronize(b) == 0
        assert self.rt.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def make_linear(N, K, device, gen, W_dtype="int4", zeros=False):
    """One synthetic quantised linear: operator + resident operands (random codes, rand*0.02 scale)."""
    cfg = bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype=W_dtype, out_dtype="float16",
                               accum_dtype="float16", group_size=GROUP, with_scaling=True,
                               with_zeros=zeros)
    op = bitblas.global_operator_cache.get(cfg)
    if op is None:
        op = bitblas.Matmul(cfg, enable_tuning=False)
        bitblas.global_operator_cache.add(cfg, op)
    qweight = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=device, generator=gen)
    scale = (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)
    out = torch.empty((1, N), dtype=torch.float16, device=device)
    return op, qweight, scale, out


def time_kernel_only(hip, device, gen, n_buf=48, reps=4):
    """Mean kernel duration of the 4096x4096 int4 GEMV from the kernel's own timestamps."""
    N = K = 4096
    op, _, _, out = make_linear(N, K, device, gen)
    bufs = [make_linear(N, K, device, gen)[1:3] for _ in range(n_buf)]   # 48 x 8.65 MB = 415 MB
    A = (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)
    stream = torch.cuda.current_stream(device).cuda_stream
    ev = [(hip.event(), hip.event()) for _ in range(n_buf)]
    durs = []
    for rep in range(reps + 1):
        for i, (qw, sc) in enumerate(bufs):
            op.lib.run_timed(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), None, None,
                             out.data_ptr(), 1, stream, ev[i][0], ev[i][1])
        torch.cuda.synchronize(device)
        if rep == 0:
            continue  # warm-up round
        durs += [hip.elapsed_ms(a, b) for a, b in ev]
    durs = np.array(durs) * 1e-3
    return float(durs.mean()), float(np.median(durs)), op.plans[1]["name"]


def time_gemm(hip, device, gen, M=4096, N=4096, K=4096, reps=20):
    """W_uint4 A_fp16 GEMM, M=4096, zeros=original (BASELINE config 3) - TFLOP/s, kernel-only."""
    try:
        cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", out_dtype="float16",
                                   accum_dtype="float16", group_size=GROUP, with_scaling=True,
                                   with_zeros=True, zeros_mode="original")
        op = bitblas.Matmul(cfg, enable_tuning=False)
        if op.plans[M]["kernel_family"] != 2:
            return None
    except Exception:
        return None
    A = (torch.rand((M, K), device=device, generator=gen) - 0.5).to(torch.float16)
    qw = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=device, generator=gen)
    sc = (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)
    zr = torch.full((N, K // GROUP), 8.0, dtype=torch.float16, device=device)
    out = torch.empty((M, N), dtype=torch.float16, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    e0, e1 = hip.event(), hip.event()
    durs = []
    for i in range(reps + 3):
        op.lib.run_timed(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), zr.data_ptr(), None,
                         out.data_ptr(), M, stream, e0, e1)
        torch.cuda.synchronize(device)
        if i >= 3:
            durs.append(hip.elapsed_ms(e0, e1) * 1e-3)
    t = float(np.mean(durs))
    return {"workload": f"W_uint4 A_fp16 GEMM M={M} N={N} K={K} g=128 zeros=original",
            "kernel": op.plans[M]["name"], "seconds": t, "tflops": 2.0 * M * N * K / t / 1e12,
            "frac_of_mfma_f16_peak": 2.0 * M * N * K / t / 1e12 / MFMA_F16_PEAK_TF}


def cpu_baseline(max_seconds=20.0):
    """The CPU oracle (numpy/torch restatement of the reference's TE definition) timed on the host
    cores: dequantise (fp16) + fp32 matmul for the M=1, N=K=4096 member.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wqaa_oracle as oracle
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    N = K = 4096
    A = (rng.random((1, K), dtype=np.float32) - 0.5).astype(np.float16)
    codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    scale = (rng.random((N, K // GROUP), dtype=np.float32) * 0.02).astype(np.float16)
    t0 = time.perf_counter()
    n = 0
    deq_t = mm_t = 0.0
    while True:
        t1 = time.perf_counter()
        Wd = oracle.dequantize_weight(codes, "int", 4, K=K, scale=scale, group_size=GROUP)
        t2 = time.perf_counter()
        out = torch.matmul(torch.from_numpy(A).float(), torch.from_numpy(Wd).float().T).half()
        t3 = time.perf_counter()
        deq_t += t2 - t1
        mm_t += t3 - t2
        n += 1
        if time.perf_counter() - t0 > max_seconds or n >= 20:
            break
    per = (deq_t + mm_t) / n
    nbytes = algorithmic_bytes(1, N, K)
    return {"value": nbytes / per / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{n} x (dequantise + fp32 matmul) of W_int4 A_fp16 M=1 N=K=4096 g=128 "
                      f"(dequant {deq_t / n * 1e3:.1f} ms + matmul {mm_t / n * 1e3:.1f} ms per pass)",
            "_check": float(out.float().abs().mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)

    # ---- resident synthetic model shard: every rank holds LAYERS x 7 linears (its N-slice) ----
    layers = []
    for _ in range(args.layers):
        layers.append([make_linear(N, K, device, gen) for (_, N, K) in LLAMA2_7B_LINEARS])
    acts = {K: (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)
            for K in (4096, 11008)}
    step_bytes = args.layers * sum(algorithmic_bytes(1, N, K) for (_, N, K) in LLAMA2_7B_LINEARS)
    launches_per_step = args.layers * len(LLAMA2_7B_LINEARS)
    stream = torch.cuda.current_stream(device).cuda_stream

    gathered = side = None
    if dist_on:
        import torch.distributed as dist
        flat_n = sum(N for (_, N, _) in LLAMA2_7B_LINEARS)
        local_out = torch.empty((args.layers, flat_n), dtype=torch.float16, device=device)
        gathered = torch.empty((world, args.layers, flat_n), dtype=torch.float16, device=device)
        side = torch.cuda.Stream(device)

    def one_step():
        for li, layer in enumerate(layers):
            off = 0
            for (op, qw, sc, out) in layer:
                A = acts[op.K]
                dst = out if not dist_on else local_out[li:li + 1, off:off + op.N]
                op.lib.run(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), None, None,
                           dst.data_ptr(), 1, stream)
                off += op.N
        if dist_on:
            import torch.distributed as dist
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(gathered, local_out)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    if dist_on:
        torch.cuda.current_stream(device).wait_stream(side)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if dist_on:
        torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        hip = Hip()
        k_mean, k_median, k_name = time_kernel_only(hip, device, gen)
        nbytes = algorithmic_bytes(1, 4096, 4096)
        achieved = nbytes / k_mean / 1e9
        value = step_bytes * args.steps * world / elapsed / 1e9
        result = {
            "metric": "achieved HBM GB/s, W_int4 A_fp16 GEMV M=1 (Llama-2-7B linear shapes, g=128); "
                      "+ TFLOP/s of the M=4096 GEMM in `gemm`",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"W_int4 A_fp16 GEMV M=1, Llama-2-7B linears {{4096,11008}}, g=128, "
                                   f"{args.layers} layers x 7 GEMV per step per GPU",
                       "launches_per_step": launches_per_step, "bytes_per_step_per_gpu": step_bytes,
                       "sharding": "column (N) shard per rank + 1 RCCL all-gather per step" if dist_on else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": k_name,
                         "bytes_per_launch": nbytes, "mean_launch_us": k_mean * 1e6,
                         "median_launch_us": k_median * 1e6,
                         "timing": "hipExtLaunchKernel start/stop events, 48 rotating 8.65 MB buffers"},
            "us_per_launch_incl_gaps": elapsed / args.steps / launches_per_step * 1e6,
        }
        gemm = time_gemm(hip, device, gen)
        if gemm:
            result["gemm"] = gemm
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline()
            cb.pop("_check", None)
            result["cpu_baseline"] = cb
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
