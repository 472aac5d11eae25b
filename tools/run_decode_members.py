"""the decode-batch members of round 5 under rocprofv3 --kernel-trace: the wave-per-fragment form (vocabulary projections), the K-sliced form and
its reduce launch, the persistent form (tools/r05_decode_stats.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, torch, bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
keys = ("kernel", "us_per_launch")
for (M, N, K) in ((8, 32000, 4096), (16, 128256, 4096), (8, 8192, 28672), (8, 22016, 4096)):
    r = bench.time_member_gemm(dev, gen, M, N, K)
    print(json.dumps({k: r.get(k) for k in keys}))
