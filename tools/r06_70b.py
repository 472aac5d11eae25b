#!/usr/bin/env python
"""Round 6: the shapes BASELINE.json's `metric` names - W_int4 A_fp16 at M = 1 and M = 4096 on the Llama-70B linears
(reference benchmark/README.md:60-62 V10-V12, :73-75 M10-M12) - through bench.py's own member timers.

    python tools/r06_70b.py [gemv] [gemm] [group]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SHAPES = [(8192, 8192), (28672, 8192), (8192, 28672), (10240, 8192)]


def main():
    what = set(sys.argv[1:]) or {"gemv", "gemm"}
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    for (N, K) in SHAPES:
        if "gemv" in what:
            r = bench.time_member_gemv(device, gen, N, K)
            print(json.dumps({"member": f"gemv_int4_n{N}k{K}", "kernel": r["kernel"], "us": round(r["us_per_launch"], 2),
                              "GBps": round(r["GBps"], 1), "frac": round(r["roofline"]["frac"], 4), "buffers": r["buffers"]}), flush=True)
        if "strict" in what:
            r = bench.time_member_gemv(device, gen, N, K, strict=True)
            print(json.dumps({"member": f"gemv_int4_n{N}k{K}_strict", "kernel": r["kernel"], "us": round(r["us_per_launch"], 2),
                              "GBps": round(r["GBps"], 1), "frac": round(r["roofline"]["frac"], 4)}), flush=True)
    if "gemm" in what:
        for (N, K) in SHAPES[:3]:
            r = bench.time_member_gemm(device, gen, 4096, N, K, n_buf=2)
            print(json.dumps({"member": f"gemm_uint4_m4096_n{N}k{K}", "kernel": r.get("kernel"), "us": round(r.get("us_per_launch", 0), 2),
                              "TFLOPs": round(r.get("TFLOPs", 0), 1), "frac": round((r.get("roofline") or {}).get("frac", 0), 4),
                              "error": r.get("error")}), flush=True)


if __name__ == "__main__":
    main()
