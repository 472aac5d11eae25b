import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (N, K) in ((4096, 11008), (22016, 4096), (4096, 4096), (8192, 8192), (28672, 8192)):
    for M in (4, 8, 16):
        bench._OPS.clear()
        r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int4")
        nbytes = N * K // 2
        print(json.dumps({"N": N, "K": K, "M": M, "plan": r["kernel"].split("_", 2)[2], "us": round(r["us_per_launch"], 2), "TBps": round(nbytes / r["us_per_launch"] / 1e6, 2)}), flush=True)
