#!/usr/bin/env python
"""Round 4: decode batches (M = 3 ... 16) on long K - the LDS-DMA decode member (blocks of 4 k-steps, drained per block) against
the direct-load member (a ring of 4 k-steps refilled as it is consumed; WQAA_GEMM_DECODE_LDS=0) and the split-K skinny member."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
KEYS = ("WQAA_GEMM_DECODE_LDS", "WQAA_GEMM_DECODE_FORCE", "WQAA_GEMM_DECODE")
for (N, K) in ((4096, 11008), (8192, 8192), (10240, 8192), (28672, 8192), (8192, 28672)):
    for M in (4, 8, 16):
        row = {"N": N, "K": K, "M": M}
        for name, env in (("default", {}), ("lds_forced", {"WQAA_GEMM_DECODE_FORCE": "1"}), ("direct_forced", {"WQAA_GEMM_DECODE_FORCE": "1", "WQAA_GEMM_DECODE_LDS": "0"}),
                          ("skinny", {"WQAA_GEMM_DECODE": "0"})):
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            bench._OPS.clear()
            r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int4")
            row[name] = (r["kernel"].split("_", 2)[2].split("_", 1)[1], round(r["us_per_launch"], 2))
        for k in KEYS:
            os.environ.pop(k, None)
        row["TBps_default"] = round(N * K / 2 / row["default"][1] / 1e6, 2)
        print(json.dumps(row), flush=True)
