#!/usr/bin/env python
"""decode batches M = 1 .. 32 of the c3 / c2 formats at N = K = 4096 (and 11008 x 4096): which member, how long (hipGraph replays)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (N, K) in ((4096, 4096), (11008, 4096)):
    for M in (1, 2, 3, 4, 5, 8, 12, 16, 32):
        bench._OPS.clear()
        r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int4") if M > 2 else None
        if r is None:
            op = bench.get_op(M, N, K)
            import bitblas_amd as bitblas
            bufs = [bench.make_linear(N, K, dev, gen)[1:3] for _ in range(32)]
            A = (torch.rand((M, K), device=dev, generator=gen) - 0.5).to(torch.float16)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            def launch_all():
                for qw, sc in bufs:
                    op(A, qw, scale=sc, output=out)
            t = bench.graph_time(dev, launch_all, len(bufs))
            print(json.dumps({"N": N, "K": K, "M": M, "plan": op.plans[M]["name"].split("_", 2)[2], "us": round(t * 1e6, 2)}), flush=True)
        else:
            print(json.dumps({"N": N, "K": K, "M": M, "plan": r["kernel"].split("_", 2)[2], "us": round(r["us_per_launch"], 2)}), flush=True)
