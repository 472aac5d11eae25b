#!/usr/bin/env python
"""tools/ab_two_libs.py: the same members timed under two builds of the library (WQAA_LIBRARY), alternating processes on one
box: bitblas_amd/libwqaa_hip_base.so (a copy of the previous build) against bitblas_amd/libwqaa_hip.so.  AB_SET=gemv (default) / gemm picks the members."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
out = {}
which = os.environ.get("AB_SET", "gemv")
if which == "head":
    for (N, K) in ((4096, 4096), (4096, 11008), (11008, 4096)):
        out[f"i4 exact {N}x{K}"] = bench.time_member_gemv(dev, gen, N, K)["us_per_launch"]
elif which == "midm":
    for M in (16, 32, 64, 128, 256, 512, 4096):
        out[f"u4 M={M}"] = bench.time_member_gemm(dev, gen, M, 4096, 4096)["us_per_launch"]
    out["u4 M=128 11008x4096"] = bench.time_member_gemm(dev, gen, 128, 11008, 4096)["us_per_launch"]
    out["i2xi8 M=128"] = bench.time_member_gemm(dev, gen, 128, 4096, 4096, W_dtype="int2", A_dtype="int8")["us_per_launch"]
elif which == "gemm":
    for (M, N, K) in ((4096, 4096, 4096), (2048, 4096, 4096), (1024, 4096, 4096), (128, 4096, 4096)):
        out[f"i2xi8 M={M}"] = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int2", A_dtype="int8")["us_per_launch"]
    out["u4 M=4096"] = bench.time_member_gemm(dev, gen, 4096, 4096, 4096)["us_per_launch"]
else:
    out["c4 step"] = bench.time_step_int2_int8(dev, gen)["us_per_step"]
    for (N, K) in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)):
        out[f"i2xi8 {N}x{K}"] = bench.time_member_dense(dev, gen, 1, N, K, kind="int2", n_buf=max(3, min(64, (640 << 20) // (N * K // 4))))["us_per_launch"]
    for (N, K) in ((4096, 4096), (12288, 4096), (22016, 4096)):
        out[f"i4 exact {N}x{K}"] = bench.time_member_gemv(dev, gen, N, K)["us_per_launch"]
print(json.dumps(out))
''' % ROOT


def main():
    libs = {"base": os.path.join(ROOT, "bitblas_amd", "libwqaa_hip_base.so"), "new": os.path.join(ROOT, "bitblas_amd", "libwqaa_hip.so")}
    res = {k: [] for k in libs}
    for rnd in range(2):
        for name, path in libs.items():
            env = dict(os.environ, WQAA_LIBRARY=path)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, "failed:", r.stderr[-400:])
                continue
            res[name].append(json.loads(line[-1]))
    keys = res["base"][0].keys() if res["base"] else []
    for k in keys:
        print(f"{k:22s} " + "  ".join(f"{n}: " + " ".join(f"{x[k]:7.2f}" for x in res[n]) for n in res))


if __name__ == "__main__":
    main()
