#!/usr/bin/env python
"""Round 4: the whole-tile form of the one-launch decode member (`xdlt`, K > 4096, M <= 8 / M <= 4) against the block-by-block
form of the same kernel (WQAA_GEMM_DECODE_LONG=0 + FORCE) and whatever the selector took before (WQAA_GEMM_DECODE_LONG=0)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
KEYS = ("WQAA_GEMM_DECODE_LONG", "WQAA_GEMM_DECODE_FORCE")
for (N, K) in ((8192, 8192), (10240, 8192), (12288, 8192), (6144, 8192), (8192, 11008), (8192, 12288), (4096, 11008)):
    for M in (3, 4, 8):
        row = {"N": N, "K": K, "M": M}
        for name, env in (("default", {}), ("before", {"WQAA_GEMM_DECODE_LONG": "0"}), ("block_by_block", {"WQAA_GEMM_DECODE_LONG": "0", "WQAA_GEMM_DECODE_FORCE": "1"})):
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            bench._OPS.clear()
            r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int4")
            row[name] = (r["kernel"].split("_", 2)[2].split("_", 1)[1], round(r["us_per_launch"], 2))
        for k in KEYS:
            os.environ.pop(k, None)
        row["TBps"] = round(N * K / 2 / row["default"][1] / 1e6, 2)
        print(json.dumps(row), flush=True)
