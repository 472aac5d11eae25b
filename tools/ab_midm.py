#!/usr/bin/env python
"""tools/ab_midm.py [M ...]: the mid-M MFMA members of bench.py (uint4 g128 + zeros, N = K = 4096 and two wider shapes),
hipGraph replays over rotating weights, microseconds per call (main kernel + reduce launch where the plan splits K)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 96, 128, 256]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        for M in Ms:
            r = [bench.time_member_gemm(dev, gen, M, N, K) for _ in range(2)]
            print(f"M={M:4d} {N}x{K} {r[0]['kernel']:48s} " + "  ".join(f"{x['us_per_launch']:7.2f}" for x in r) + " us")


if __name__ == "__main__":
    main()
