#!/usr/bin/env python
"""tools/r05_ab_wpf.py: same-process A/B of the wave-per-fragment decode form (plan suffix `xdlw`, member 213: WQAA_GEMM_DECODE_LONG=4 takes it
wherever it fits) against the selector's members, uint4 g128 + zeros, hipGraph replays over rotating weights, two alternating repeats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ARMS = (("wpf", {"WQAA_GEMM_DECODE_LONG": "4"}), ("sel", {})) if len(sys.argv) < 2 else (("dflt", {}), ("r4", {"WQAA_GEMM_DECODE_LONG": "2"}))


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    shapes = []
    table = ((22016, 4096), (11008, 4096), (12288, 4096), (32000, 4096), (4096, 4096), (28672, 4096), (16384, 8192), (8192, 8192))
    if len(sys.argv) > 1:      # second run: the selector's rule against round 4's selector
        table = ((32000, 4096), (128256, 4096), (28672, 4096), (16384, 8192), (14336, 8192), (24576, 4096), (65536, 2048))
    for (N, K) in table:
        for M in (4, 8, 16):
            if K > 4096 and M > 8:
                continue
            shapes.append((M, N, K))
    for (M, N, K) in shapes:
        row = []
        for rep in range(2):
            for arm, env in ARMS:
                os.environ.pop("WQAA_GEMM_DECODE_LONG", None)
                os.environ.update(env)
                bench._OPS.clear()
                r = bench.time_member_gemm(dev, gen, M, N, K)
                row.append((arm, r.get("kernel", "?").split("_")[-1], r.get("us_per_launch", float("nan"))))
        os.environ.pop("WQAA_GEMM_DECODE_LONG", None)
        w_mb = N * K / 2 / 1e6
        best = min(t for a, _, t in row if a in ("wpf", "dflt"))
        print(f"M={M:3d} {N}x{K} ({w_mb:6.1f} MB, {w_mb / best / 1e3:5.2f} TB/s)  " + "  ".join(f"{a}:{k} {t:7.2f}" for a, k, t in row), flush=True)


if __name__ == "__main__":
    main()
