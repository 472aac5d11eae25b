#!/usr/bin/env python
"""tools/blaslt_probe.py: what the vendor library (hipBLASLt through torch) does on the plain dense GEMMs of BASELINE c5
(e4m3 x e4m3, M = 4096, Llama-3-70B shapes) and on dense float16, next to this library's members - same operands, hipGraph
replays.  TFLOP/s."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402


def graph_us(fn, n=4, replays=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(ts))


def sweep_m(dev):
    one = torch.ones((), device=dev, dtype=torch.float32)
    for (N, K) in ((8192, 8192), (1280, 8192), (8192, 28672)):
        W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.float8_e4m3fn)
        for M in (8, 16, 64, 128, 256, 512, 1024, 2048):
            A = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.float8_e4m3fn)
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            t_own, n_own, t_def, n_def, op = both(cfg, M, lambda o: graph_us(lambda: o(A, W, output=out), n=8))
            print(f"e4m3 N={N} K={K} M={M:5d}: own member {t_own:8.1f} us ({n_own}) | product default {t_def:8.1f} us ({n_def})")
    for (N, K) in ((4096, 4096),):
        W = (torch.rand((N, K), device=dev) - 0.5).half()
        for M in (16, 64, 128, 256, 512, 1024, 2048):
            A = (torch.rand((M, K), device=dev) - 0.5).half()
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="float16", accum_dtype="float32", out_dtype="float16")
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            t_own, n_own, t_def, n_def, op = both(cfg, M, lambda o: graph_us(lambda: o(A, W, output=out), n=8))
            print(f"f16  N={N} K={K} M={M:5d}: own member {t_own:8.1f} us ({n_own}) | product default {t_def:8.1f} us ({n_def})")


def both(cfg, M, run):
    """time the operator as the own member (WQAA_DENSE_LIB=0, a plan-time switch) and as the product default"""
    os.environ["WQAA_DENSE_LIB"] = "0"
    own = bitblas.Matmul(cfg, enable_tuning=False)
    t_own, name_own = run(own), own.plans[M]["name"].split("_", 2)[2]
    del os.environ["WQAA_DENSE_LIB"]
    dflt = bitblas.Matmul(cfg, enable_tuning=False)
    t_def, name_def = run(dflt), dflt.plans[M]["name"].split("_", 2)[2]
    return t_own, name_own, t_def, name_def, dflt


def main():
    dev = torch.device("cuda", 0)
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        return sweep_m(dev)
    M = 4096
    one = torch.ones((), device=dev, dtype=torch.float32)
    for (N, K) in ((8192, 8192), (8192, 28672), (10240, 8192), (28672, 8192)):
        A = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.float8_e4m3fn)
        W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.float8_e4m3fn)
        cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        t_own, n_own, t_def, n_def, op = both(cfg, M, lambda o: graph_us(lambda: o(A, W, output=out)))
        ref = None
        try:
            t_lt = graph_us(lambda: torch._scaled_mm(A, W.t(), scale_a=one, scale_b=one, out_dtype=torch.float16))
            ref = torch._scaled_mm(A, W.t(), scale_a=one, scale_b=one, out_dtype=torch.float16)
        except Exception as e:  # noqa: BLE001
            t_lt = float("nan")
            print("scaled_mm failed:", type(e).__name__, str(e)[:200])
        fl = 2.0 * M * N * K
        diff = float((ref.float() - out.float()).abs().max() / out.float().abs().max()) if ref is not None else float("nan")
        print(f"e4m3 M={M} N={N} K={K}: own member ({n_own}) {t_own:8.1f} us {fl / t_own / 1e6:7.0f} TF | product default ({n_def}) {t_def:8.1f} us "
              f"{fl / t_def / 1e6:7.0f} TF | torch._scaled_mm {t_lt:8.1f} us {fl / t_lt / 1e6:7.0f} TF | default vs torch max rel diff {diff:.2e}")
    for (N, K) in ((4096, 4096), (8192, 8192)):
        A = (torch.rand((M, K), device=dev) - 0.5).half()
        W = (torch.rand((N, K), device=dev) - 0.5).half()
        cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="float16", accum_dtype="float32", out_dtype="float16")
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        t_own, n_own, t_def, n_def, op = both(cfg, M, lambda o: graph_us(lambda: o(A, W, output=out)))
        t_lt = graph_us(lambda: torch.matmul(A, W.t()))
        fl = 2.0 * M * N * K
        print(f"f16  M={M} N={N} K={K}: own member ({n_own}) {t_own:8.1f} us {fl / t_own / 1e6:7.0f} TF | product default ({n_def}) {t_def:8.1f} us "
              f"{fl / t_def / 1e6:7.0f} TF | torch.matmul {t_lt:8.1f} us {fl / t_lt / 1e6:7.0f} TF")


if __name__ == "__main__":
    main()
