#!/usr/bin/env python
"""Round 4: the automatic two-pass form WITHOUT the vendor library (B_decode -> this library's dense 16-bit ping-pong member) against
the fused lockstep member it replaces (WQAA_TWO_PASS_AUTO=0), for the float16 / bfloat16 formats that have no fused ping-pong
member, hipGraph replays, same process."""
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


def time_one(M, N, K, a, w, **kw):
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=a, W_dtype=w, out_dtype=a, accum_dtype="float32", **kw)
    op = bitblas.Matmul(cfg, enable_tuning=False)
    tdt = torch.float16 if a == "float16" else torch.bfloat16
    A = (torch.rand((M, K), device=dev, generator=gen) - 0.5).to(tdt)
    nb = 4
    Ws = [torch.randint(-128, 128, (N, K * op.bit // 8), dtype=torch.int8, device=dev, generator=gen) for _ in range(nb)]
    if w == "e4m3_float8":
        Ws = [(torch.rand((N, K), device=dev, generator=gen) * 2 - 1).to(torch.float8_e4m3fn) for _ in range(nb)]
    g = kw.get("group_size", -1)
    sc = (torch.rand((N, K // (g if g > 0 else K)), device=dev, generator=gen) * 0.02).to(tdt) if kw.get("with_scaling") else None
    out = torch.empty((M, N), dtype=tdt, device=dev)

    def launch_all():
        for W in Ws:
            op(A, W, scale=sc, output=out)

    t = bench.graph_time(dev, launch_all, nb)
    return op.plans[M]["name"].split("_", 2)[2], round(t * 1e6, 1)


def time_i8(M, N, K, w):
    cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int8", W_dtype=w, out_dtype="int32", accum_dtype="int32")
    op = bitblas.Matmul(cfg, enable_tuning=False)
    A = torch.randint(-128, 128, (M, K), device=dev, dtype=torch.int8, generator=gen)
    Ws = [torch.randint(-128, 128, (N, K * op.bit // 8), dtype=torch.int8, device=dev, generator=gen) for _ in range(4)]
    out = torch.empty((M, N), dtype=torch.int32, device=dev)

    def launch_all():
        for W in Ws:
            op(A, W, output=out)

    t = bench.graph_time(dev, launch_all, 4)
    return op.plans[M]["name"].split("_", 2)[2], round(t * 1e6, 1)


for w in ("int8", "uint4", "int1"):
    for (M, N, K) in ((2048, 4096, 4096), (4096, 4096, 4096), (4096, 11008, 4096)):
        row = {"a": "int8", "w": w, "shape": [M, N, K]}
        for name, env in (("new", {}), ("lockstep", {"WQAA_TWO_PASS_AUTO": "0", "WQAA_GEMM_PP": "0"})):
            for k in ("WQAA_TWO_PASS_AUTO", "WQAA_GEMM_PP"):
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                row[name] = time_i8(M, N, K, w)
            except Exception as e:  # noqa: BLE001
                row[name] = f"{type(e).__name__}: {e}"[:120]
        for k in ("WQAA_TWO_PASS_AUTO", "WQAA_GEMM_PP"):
            os.environ.pop(k, None)
        print(json.dumps(row), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "--int8-only":
    sys.exit(0)

for (a, w, kw) in (("float16", "int8", {}), ("float16", "uint2", dict(group_size=128, with_scaling=True)), ("float16", "e4m3_float8", {}),
                   ("float16", "uint1", {}), ("bfloat16", "int8", dict(group_size=128, with_scaling=True))):
    for (M, N, K) in ((1024, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (4096, 11008, 4096)):
        row = {"a": a, "w": w, "shape": [M, N, K]}
        for name, env in (("auto_two_pass", {}), ("fused_lockstep", {"WQAA_TWO_PASS_AUTO": "0"})):
            os.environ.pop("WQAA_TWO_PASS_AUTO", None)
            os.environ.update(env)
            try:
                row[name] = time_one(M, N, K, a, w, **kw)
            except Exception as e:  # noqa: BLE001
                row[name] = f"{type(e).__name__}: {e}"[:120]
        os.environ.pop("WQAA_TWO_PASS_AUTO", None)
        print(json.dumps(row), flush=True)
