#!/usr/bin/env python
"""Round 4: the persistent form of the one-launch decode member against what the selector chose before (WQAA_GEMM_DECODE_PERSIST=0:
the split-K skinny member + reduce, or one fragment per workgroup over partial rounds), M = 3 ... 16, hipGraph replays."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (N, K) in ((11008, 4096), (12288, 4096), (22016, 4096), (8192, 4096), (5120, 4096)):
    for M in (3, 4, 8, 16):
        row = {"N": N, "K": K, "M": M}
        for name, env in (("persistent", {}), ("before", {"WQAA_GEMM_DECODE_PERSIST": "0"}), ("one_fragment_each", {"WQAA_GEMM_DECODE_PERSIST": "0", "WQAA_GEMM_DECODE_FORCE": "1"})):
            for k in ("WQAA_GEMM_DECODE_PERSIST", "WQAA_GEMM_DECODE_FORCE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            bench._OPS.clear()
            r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int4")
            row[name] = (r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 2))
        for k in ("WQAA_GEMM_DECODE_PERSIST", "WQAA_GEMM_DECODE_FORCE"):
            os.environ.pop(k, None)
        print(json.dumps(row), flush=True)
for M in (4, 16):
    row = {"i2xi8 N": 11008, "M": M}
    for name, env in (("persistent", {}), ("before", {"WQAA_GEMM_DECODE_PERSIST": "0"})):
        os.environ.pop("WQAA_GEMM_DECODE_PERSIST", None); os.environ.update(env)
        bench._OPS.clear()
        r = bench.time_member_gemm(dev, gen, M, 11008, 4096, W_dtype="int2", A_dtype="int8")
        row[name] = (r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 2))
    os.environ.pop("WQAA_GEMM_DECODE_PERSIST", None)
    print(json.dumps(row), flush=True)
