#!/usr/bin/env python
"""tools/ab_midm_knobs.py "M N K" ... -- "VAR=v ..." ...: mid-M MFMA members (uint4 g128 + zeros) under selector tuning
variables (WQAA_GEMM_KSPLIT / _MF / _SKINNY_MAXM / ...), same process, two rounds, microseconds per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    shapes = [tuple(int(x) for x in a.split()) for a in argv[:cut]]
    combos = [""] + argv[cut + 1:]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (M, N, K) in shapes:
        res = {}
        for rnd in range(2):
            for combo in combos:
                kv = dict(x.split("=") for x in combo.split()) if combo else {}
                os.environ.update(kv)
                op = bench.get_op(M, N, K, W_dtype="uint4", zeros=True)
                op.lib.plan(M)                      # re-plan under the variables
                op.plans[M] = op.lib.plan(M)
                r = bench.time_member_gemm(dev, gen, M, N, K)
                res.setdefault(combo, [r["kernel"].split("_", 2)[2] + f" s{op.plans[M]['split_k']}"]).append(r["us_per_launch"])
                for k in kv:
                    del os.environ[k]
        op.plans[M] = op.lib.plan(M)
        for combo, v in res.items():
            print(f"M={M} {N}x{K} {combo or 'default':40s} {v[0]:36s} " + "  ".join(f"{x:7.2f}" for x in v[1:]))


if __name__ == "__main__":
    main()
