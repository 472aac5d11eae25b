import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, torch, bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for kw in (dict(), dict(bitnet=True)):
    r = bench.time_member_gemm(dev, gen, 4096, W_dtype="int2", A_dtype="int8", **kw)
    print(json.dumps({k: r[k] for k in ("kernel", "us_per_launch", "TFLOPs", "frac_of_mfma_peak")}))
r = bench.time_member_gemm(dev, gen, 4096)
print(json.dumps({k: r[k] for k in ("kernel", "us_per_launch", "TFLOPs", "frac_of_mfma_peak")}))
