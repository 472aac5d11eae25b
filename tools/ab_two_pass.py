#!/usr/bin/env python
"""tools/ab_two_pass.py: fused MFMA member against the tuned choice (`Matmul.hardware_aware_finetune`: two-pass member =
B_decode to a scratch + the vendor GEMM where it measures faster) on bench.py's operands, hipGraph replays."""
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
    for (M, N, K, wd, ad) in ((4096, 4096, 4096, "uint4", "float16"), (1024, 4096, 4096, "uint4", "float16"), (4096, 11008, 4096, "uint4", "float16"),
                              (4096, 4096, 11008, "uint4", "float16"), (4096, 8192, 8192, "uint4", "float16"), (4096, 4096, 4096, "int2", "int8"),
                              (4096, 8192, 8192, "int2", "int8")):
        a = bench.time_member_gemm(dev, gen, M, N, K, W_dtype=wd, A_dtype=ad, n_buf=4)
        b = bench.time_member_gemm(dev, gen, M, N, K, W_dtype=wd, A_dtype=ad, n_buf=4, tuned=True)
        print(f"M={M} N={N} K={K} W_{wd} A_{ad}: fused {a['us_per_launch']:8.1f} us {a['TFLOPs']:7.0f} T | tuned {b['us_per_launch']:8.1f} us "
              f"{b['TFLOPs']:7.0f} T [{b['kernel'].split('_', 2)[2]}] {b.get('tuning')}")


def dense():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (N, K) in ((8192, 8192), (8192, 28672), (10240, 8192), (28672, 8192)):
        a = bench.time_member_dense(dev, gen, 4096, N, K, n_buf=2)
        b = bench.time_member_dense(dev, gen, 4096, N, K, n_buf=2, tuned=True)
        print(f"e4m3 M=4096 N={N} K={K}: heuristic {a['us_per_launch']:8.1f} us {a['TFLOPs']:7.0f} T | tuned {b['us_per_launch']:8.1f} us {b['TFLOPs']:7.0f} T")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dense":
        dense()
        sys.exit(0)
    main()
