#!/usr/bin/env python
"""tools/r05_ab_mid.py: same-process A/B of the mid-M member (plan suffix `xmk`, csrc/wqaa_gemm_mid_kernel.h) against the members it
stands in for (WQAA_GEMM_MID=0), uint4 g128 + zeros, hipGraph replays over rotating weights (bench.py's member harness), two
alternating repeats per arm.  WQAA_GEMM_MID_MINM / _MAXM / _ROUNDS open the selector's fences for the shapes outside them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

OPEN = {"WQAA_GEMM_MID_MINM": "3", "WQAA_GEMM_MID_MAXM": "512", "WQAA_GEMM_MID_ROUNDS": "4"}
ARMS = (("mid", dict(OPEN)),                                            # two-launch seam (default)
        ("inl", dict(OPEN, WQAA_GEMM_MID_SEAM="1")),                    # the slices meet inside the launch
        ("old", {"WQAA_GEMM_MID": "0"}))

def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    shapes = [(M, 4096, 4096) for M in (16, 17, 32, 48, 64, 96, 128, 192, 256)]
    shapes += [(128, 2048, 4096), (64, 4096, 8192), (32, 4096, 8192), (128, 4096, 2048), (128, 8192, 4096), (64, 8192, 4096), (128, 11008, 4096), (64, 11008, 4096)]
    for (M, N, K) in shapes:
        row = []
        for rep in range(2):
            for arm, env in ARMS:
                if arm == "inl" and rep:
                    continue
                for k in list(os.environ):
                    if k.startswith("WQAA_GEMM_MID"):
                        del os.environ[k]
                os.environ.update(env)
                bench._OPS.clear()          # (bench.get_op caches operators: the plan is made when the operator is)
                r = bench.time_member_gemm(dev, gen, M, N, K)
                row.append((arm, r.get("kernel", "?").split("_")[-1], r.get("us_per_launch", float("nan"))))
        print(f"M={M:4d} {N}x{K}  " + "  ".join(f"{a}:{k} {t:7.2f}" for a, k, t in row), flush=True)


if __name__ == "__main__":
    main()
