#!/usr/bin/env python
"""Round-4 same-box A/Bs of the large-M members (one process, hipGraph replays, bench.py's own timing functions):
 (1) bench.graph_time with 0 / 25 / 100 ms of untimed replays in front (the clocks the chip settles at under load);
 (2) the BitNet epilogue's two IEEE divisions: shared-divisor form (ExactDiv) against fp64 (WQAA_GEMM_WS_POLICY bit 5);
 (3) the remainder of a partial round as a second launch of the 128-row tile (WQAA_GEMM_PP_TAIL=0 / default) on the shapes
     VERDICT r03 item 6 names, with the forced single-tile choices next to it."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1)


def one(**kw):
    bench._OPS.clear()
    r = bench.time_member_gemm(dev, gen, **kw)
    return (r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 2))


print("== (1) untimed replays in front of the timed ones")
for warm in (0.0, 25.0, 100.0, 0.0, 25.0):
    bench.GRAPH_WARM_MS = warm
    print(json.dumps({"warm_ms": warm, "u4": one(M=4096), "i2": one(M=4096, W_dtype="int2", A_dtype="int8"),
                      "i2_bitnet": one(M=4096, W_dtype="int2", A_dtype="int8", bitnet=True)}), flush=True)
bench.GRAPH_WARM_MS = 25.0
print("== (2) BitNet epilogue divisions")
for rep in range(2):
    row = {}
    for name, pol in (("exactdiv", None), ("fp64", "51")):
        if pol is None:
            os.environ.pop("WQAA_GEMM_WS_POLICY", None)
        else:
            os.environ["WQAA_GEMM_WS_POLICY"] = pol
        row[name] = one(M=4096, W_dtype="int2", A_dtype="int8", bitnet=True)
    os.environ.pop("WQAA_GEMM_WS_POLICY", None)
    row["int32_out"] = one(M=4096, W_dtype="int2", A_dtype="int8")
    print(json.dumps(row), flush=True)
print("== (3) partial rounds")
for (M, N, K) in ((2048, 11008, 4096), (4096, 11008, 4096), (1536, 11008, 4096), (2048, 4096, 11008), (3072, 11008, 4096)):
    row = {}
    for name, env in (("default", {}), ("no_tail", {"WQAA_GEMM_PP_TAIL": "0"}), ("force256", {"WQAA_GEMM_PP_BM": "256"}), ("force128", {"WQAA_GEMM_PP_BM": "128"})):
        for k in ("WQAA_GEMM_PP_TAIL", "WQAA_GEMM_PP_BM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        row[name] = one(M=M, N=N, K=K)
    for k in ("WQAA_GEMM_PP_TAIL", "WQAA_GEMM_PP_BM"):
        os.environ.pop(k, None)
    print(json.dumps({"shape": [M, N, K], **row}), flush=True)
