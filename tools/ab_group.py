#!/usr/bin/env python
"""tools/ab_group.py - same-process A/B of wqaa_matmul_group on the headline shapes (M = 1, int4 g128, K = 4096):
the members one launch each, the group as one launch, and the concatenated operator (N = sum) as the reference's
fuse_qkv / fuse_gateup would run it; optionally with the grid cap lifted (WQAA_GEMVX_GRID).  hipGraph replays over
rotating weight sets (> Infinity Cache), microseconds per layer-group."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    strict = os.environ.get("AB_STRICT", "0") == "1"
    for (name, Ns, K) in (("qkv", [4096] * 3, 4096), ("gate_up", [11008] * 2, 4096), ("gqa_qkv", [4096, 1024, 1024], 4096),
                          ("qkv_70b_shard", [1024, 128, 128], 8192)):
        total = sum(Ns)
        nset = max(4, (640 << 20) // (total * K // 2))
        A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
        sets = []
        for _ in range(nset):
            sets.append([bench.make_linear(N, K, dev, gen)[1:3] for N in Ns])
        ops = [bench.get_op(1, N, K, strict=strict) for N in Ns]
        merged = bench.get_op(1, total, K, strict=strict)
        mw = [(torch.cat([w for w, _ in s]), torch.cat([sc for _, sc in s])) for s in sets]
        outs = [torch.empty((1, N), dtype=torch.float16, device=dev) for N in Ns]
        mout = torch.empty((1, total), dtype=torch.float16, device=dev)

        def singles():
            st = torch.cuda.current_stream(dev).cuda_stream
            for s in sets:
                for op, (w, sc), o in zip(ops, s, outs):
                    op.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr(), None, None, o.data_ptr(), 1, st)

        def grouped():
            for s in sets:
                bitblas.matmul_group(ops, A, s, outputs=outs)

        def concat():
            st = torch.cuda.current_stream(dev).cuda_stream
            for (w, sc) in mw:
                merged.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr(), None, None, mout.data_ptr(), 1, st)

        nbytes = sum(bench.algorithmic_bytes(1, N, K) for N in Ns)
        row = {}
        for env in ({}, {"WQAA_GEMVX_GRID": "100000"}):
            for k, v in env.items():
                os.environ[k] = v
            plan = bitblas.group_plan(ops, 1)       # bumps the plan epoch: the tuning variables are re-read
            merged.lib.plan(1)
            tag = "uncapped" if env else "default"
            for label, fn in (("singles", singles), ("group", grouped), ("concat", concat)):
                t = bench.graph_time(dev, fn, nset, replays=7)
                row[f"{label}_{tag}"] = t * 1e6
            for k in env:
                del os.environ[k]
            bitblas.group_plan(ops, 1)
            if not env:
                pname = plan["plan"]["name"] if plan["plan"] else "unfused"
        print(f"{name:14s} N={Ns} K={K} {nbytes / 1e6:6.1f} MB  " + "  ".join(f"{k} {v:6.2f}" for k, v in row.items()) +
              f"  | group {nbytes / row['group_default'] / 1e3:6.0f} GB/s vs singles {nbytes / row['singles_default'] / 1e3:6.0f}  [{pname}]")


if __name__ == "__main__":
    main()
