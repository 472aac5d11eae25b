import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, torch, bench
# the decode-batch members of bench.py (M = 8; persistent `xdlp` and whole-tile `xdlt` forms) for a rocprofv3 --kernel-trace --stats run
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (N, K) in ((11008, 4096), (22016, 4096), (8192, 8192)):
    r = bench.time_member_gemm(dev, gen, 8, N, K)
    print(json.dumps({k: r[k] for k in ("kernel", "us_per_launch", "GBps_algorithmic")}))
