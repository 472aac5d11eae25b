#!/usr/bin/env python
"""tools/r05_ab_kslice.py: same-process A/B of the K-sliced decode form (plan suffix `xdlk`, csrc/wqaa_gemm_kernel.h member 212) against the
members it stands in for (WQAA_GEMM_DECODE_LONG=2: the round-4 selector), uint4 g128 + zeros, hipGraph replays over rotating weights
(bench.py's member harness), two alternating repeats per arm.  WQAA_GEMM_DECODE_LONG=3 takes the form wherever it fits."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ARMS = (("ksl", {"WQAA_GEMM_DECODE_LONG": "3"}), ("dflt", {}), ("old", {"WQAA_GEMM_DECODE_LONG": "2"}))


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    shapes = []
    for (N, K) in ((4096, 11008), (8192, 28672), (4096, 14336), (8192, 8192), (12288, 8192), (4096, 8192), (5120, 13824), (11008, 8192)):
        for M in (4, 8, 16):
            shapes.append((M, N, K))
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        shapes = [(8, 4096, 11008), (8, 8192, 28672), (16, 8192, 28672), (4, 8192, 28672), (16, 12288, 8192), (8, 12288, 8192), (9, 8192, 8192)]
    for (M, N, K) in shapes:
        row = []
        for rep in range(1):
            for arm, env in ARMS:
                for k in ("WQAA_GEMM_DECODE_LONG",):
                    os.environ.pop(k, None)
                os.environ.update(env)
                bench._OPS.clear()          # (bench.get_op caches operators: the plan is made when the operator is)
                r = bench.time_member_gemm(dev, gen, M, N, K)
                row.append((arm, r.get("kernel", "?").split("_")[-1], r.get("us_per_launch", float("nan"))))
        os.environ.pop("WQAA_GEMM_DECODE_LONG", None)
        w_mb = N * K / 2 / 1e6
        best = min(t for a, _, t in row if a != "old")
        print(f"M={M:3d} {N}x{K} ({w_mb:6.1f} MB, {w_mb / best / 1e3:5.2f} TB/s)  " + "  ".join(f"{a}:{k} {t:7.2f}" for a, k, t in row), flush=True)


if __name__ == "__main__":
    main()
