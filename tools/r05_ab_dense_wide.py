#!/usr/bin/env python
"""tools/r05_ab_dense_wide.py: same-process A/B of the dense 256 x 256 tile's two wave grids - 2 x 4 (wq_gemm_pp8w_kernel, round 5)
against 1 x 8 (wq_gemm_pp8_kernel, WQAA_GEMM_PP8_WIDE=0) - on bench.py's dense members: float16 / int8 4096^3, e4m3 on the Llama-3-70B
linears (BASELINE c5), plus the vendor yardstick on the float16 one.  hipGraph replays, two alternating repeats per arm."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    cases = [("f16 4096^3", dict(M=4096, N=4096, K=4096, kind="f16", n_buf=4)), ("int8 4096^3", dict(M=4096, N=4096, K=4096, kind="int8", n_buf=4)),
             ("e4m3 o 8192x8192", dict(M=4096, N=8192, K=8192, n_buf=4)), ("e4m3 qkv 10240x8192", dict(M=4096, N=10240, K=8192, n_buf=4)),
             ("e4m3 down 8192x28672", dict(M=4096, N=8192, K=28672, n_buf=2)), ("e4m3 gate 28672x8192", dict(M=4096, N=28672, K=8192, n_buf=2)),
             ("f16 8192x4096^2", dict(M=8192, N=4096, K=4096, kind="f16", n_buf=4)), ("f16 2048x4096^2", dict(M=2048, N=4096, K=4096, kind="f16", n_buf=4))]
    for name, kw in cases:
        row = []
        for rep in range(2):
            for arm in ("1", "0"):
                os.environ["WQAA_GEMM_PP8_WIDE"] = arm
                r = bench.time_member_dense(dev, gen, kw["M"], kw["N"], kw["K"], kind=kw.get("kind", "fp8"), n_buf=kw["n_buf"])
                row.append((("2x4" if arm == "1" else "1x8"), r.get("us_per_launch", float("nan")), (r.get("roofline") or {}).get("frac", float("nan"))))
        print(f"{name:24s} " + "  ".join(f"{a} {t:8.2f} us ({f:.3f})" for a, t, f in row), flush=True)
    os.environ.pop("WQAA_GEMM_PP8_WIDE", None)
    r = bench.time_member_dense(dev, gen, 4096, 4096, 4096, kind="f16", n_buf=4, vendor=True, tuned=True)
    print(f"f16 4096^3 vendor (tuned) {r.get('us_per_launch', float('nan')):8.2f} us", flush=True)


if __name__ == "__main__":
    main()
