import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
# WQAA_GEMM_WS_POLICY: 3 = partial sums write-through only, 19 = + output tiles write-through
for (M, N, K) in ((512, 4096, 4096), (1024, 4096, 4096), (512, 11008, 4096), (4096, 4096, 4096), (2048, 4096, 4096)):
    row = []
    for rnd in range(3):
        for pol in ("3", "19"):
            os.environ["WQAA_GEMM_WS_POLICY"] = pol
            bench._OPS.clear()
            r = bench.time_member_gemm(dev, gen, M, N, K)
            row.append((pol, r["us_per_launch"], r["kernel"].split("_")[-1]))
    print(f"M={M} {N}x{K} {row[0][2]}: " + "  ".join(f"p{p} {u:6.2f}" for p, u, _ in row))
