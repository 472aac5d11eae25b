#!/usr/bin/env python
"""tools/ab_pp_tile.py [kind] "M N K" ...: the tile the selector picks for a large-M GEMM against each forced choice
(WQAA_GEMM_PP_BM = 256 / 128 / 0: ping-pong 256 x 256, ping-pong 128 x 256, lockstep members), same process, hipGraph
replays over rotating weights.  kind: u4 (uint4 g128 + zeros x fp16, default), i2 (int2 x int8), f8 (e4m3 x e4m3), f16 / i8 (dense
float16 x float16 / int8 x int8, round 4)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_one(dev, gen, kind, M, N, K):
    bench._OPS.clear()                      # a fresh operator: planned (and named) under the variable just set
    if kind == "f8":
        r = bench.time_member_dense(dev, gen, M, N, K, kind="fp8")
    elif kind in ("f16", "i8"):
        r = bench.time_member_dense(dev, gen, M, N, K, kind="f16" if kind == "f16" else "int8", n_buf=4)
    elif kind == "i2":
        r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int2", A_dtype="int8")
    else:
        r = bench.time_member_gemm(dev, gen, M, N, K)
    if r is None or "error" in r:
        return ("refused", float("nan"))
    return (r["kernel"].split("_", 2)[2], r["us_per_launch"])


def main():
    argv = sys.argv[1:]
    kind = "u4"
    if argv and argv[0] in ("u4", "i2", "f8", "f16", "i8"):
        kind = argv.pop(0)
    shapes = [tuple(int(x) for x in a.split()) for a in argv]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (M, N, K) in shapes:
        row = []
        for force in (None, "256", "128", "0") + (("128x128",) if kind == "f8" else ()):
            os.environ.pop("WQAA_GEMM_PP_BN", None)
            if force is None:
                os.environ.pop("WQAA_GEMM_PP_BM", None)
            elif force == "128x128":
                os.environ["WQAA_GEMM_PP_BM"] = "128"
                os.environ["WQAA_GEMM_PP_BN"] = "128"
            else:
                os.environ["WQAA_GEMM_PP_BM"] = force
            name, us = time_one(dev, gen, kind, M, N, K)
            row.append((force or "selector", name, us))
        os.environ.pop("WQAA_GEMM_PP_BM", None)
        os.environ.pop("WQAA_GEMM_PP_BN", None)
        best = min(x[2] for x in row[1:] if x[2] == x[2])
        print(f"{kind} M={M} N={N} K={K}: " + " | ".join(f"{f}: {n.split('_')[-1]} {u:7.1f}" for f, n, u in row) +
              f" | selector/best = {row[0][2] / best:.3f}")


if __name__ == "__main__":
    main()
