#!/usr/bin/env python
"""Round 4: the dense float16 member on the ping-pong skeleton (csrc/wqaa_gemm_pp_kernel.h, PP8Policy<2, 2>) against the lockstep
member it replaces (WQAA_GEMM_PP=0) and the vendor library (WQAA_DENSE_LIB=1), and the B_decode pass (wqaa_dequantize) that
would sit in front of it in a two-pass plan - same process, hipGraph replays (bench.py's timing functions)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
for (M, N, K) in ((4096, 4096, 4096), (2048, 4096, 4096), (4096, 11008, 4096), (8192, 4096, 4096)):
    row = {"shape": [M, N, K]}
    for name, env in (("own_pp", {}), ("own_lockstep", {"WQAA_GEMM_PP": "0"}), ("vendor", {"WQAA_DENSE_LIB": "1"})):
        for k in ("WQAA_GEMM_PP", "WQAA_DENSE_LIB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        bench._OPS.clear()
        r = bench.time_member_dense(dev, gen, M, N, K, kind="f16", n_buf=4, tuned=(name == "vendor"))
        row[name] = (r.get("kernel", "?").split("_", 2)[-1], round(r.get("us_per_launch", float("nan")), 1), round(r.get("frac_of_mfma_peak", 0.0), 3)) if "error" not in r else r
    for k in ("WQAA_GEMM_PP", "WQAA_DENSE_LIB"):
        os.environ.pop(k, None)
    # the B_decode pass of the same weight shape (uint4 g128 + zeros -> float16)
    op = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True,
                                             zeros_mode="original"), enable_tuning=False)
    qw = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=gen)
    sc = (torch.rand((N, K // 128), device=dev, generator=gen) * 0.02).to(torch.float16)
    zr = torch.full((N, K // 128), 8.0, dtype=torch.float16, device=dev)
    outs = [torch.empty((N, K), dtype=torch.float16, device=dev) for _ in range(4)]

    def launch_all():
        for o in outs:
            op.dequantize_weight(qw, scale=sc, zeros=zr, out=o)

    try:
        row["b_decode_us"] = round(bench.graph_time(dev, launch_all, len(outs)) * 1e6, 1)
    except Exception as e:  # noqa: BLE001
        row["b_decode_us"] = f"{type(e).__name__}: {e}"
    r = bench.time_member_gemm(dev, gen, M, N, K)
    row["fused_uint4"] = (r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 1))
    print(json.dumps(row), flush=True)
