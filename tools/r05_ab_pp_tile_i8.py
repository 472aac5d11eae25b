#!/usr/bin/env python
"""tools/r05_ab_pp_tile_i8.py: the ping-pong tile shape of the int8 members (WQAA_GEMM_PP_BM / _BN force it) against the selector's choice:
int2 x int8 (int32 out, BitNet float16 out) and dense int8 at BASELINE c4's 4096^3 and two larger shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ARMS = (("sel", {}), ("256x256", {"WQAA_GEMM_PP_BM": "256", "WQAA_GEMM_PP_BN": "256"}), ("128x256", {"WQAA_GEMM_PP_BM": "128", "WQAA_GEMM_PP_BN": "256"}),
        ("128x128", {"WQAA_GEMM_PP_BM": "128", "WQAA_GEMM_PP_BN": "128"}))


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (M, N, K, kw, tag) in ((4096, 4096, 4096, dict(W_dtype="int2", A_dtype="int8"), "i2xi8 i32"), (4096, 4096, 4096, dict(W_dtype="int2", A_dtype="int8", bitnet=True), "i2xi8 bitnet"),
                               (4096, 8192, 4096, dict(W_dtype="int2", A_dtype="int8"), "i2xi8 i32"), (4096, 4096, 4096, dict(), "u4xf16"), (2048, 4096, 4096, dict(W_dtype="int2", A_dtype="int8"), "i2xi8 i32")):
        row = []
        for rep in range(2):
            for arm, env in ARMS:
                for k in ("WQAA_GEMM_PP_BM", "WQAA_GEMM_PP_BN"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                bench._OPS.clear()
                r = bench.time_member_gemm(dev, gen, M, N, K, **kw)
                row.append((arm, (r or {}).get("kernel", "?").split("_")[-1], (r or {}).get("us_per_launch", float("nan"))))
        for k in ("WQAA_GEMM_PP_BM", "WQAA_GEMM_PP_BN"):
            os.environ.pop(k, None)
        print(f"{tag:13s} M={M} {N}x{K}  " + "  ".join(f"{a}:{k} {t:7.2f}" for a, k, t in row), flush=True)


if __name__ == "__main__":
    main()
