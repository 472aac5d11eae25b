#!/usr/bin/env python
"""tools/ab_two_builds.py <old.so> [rounds]: the decode members of two BUILDS of the library on one box, interleaved
processes (each process loads one build through WQAA_LIBRARY): the headline step, the chained step with the layer's ops, the four
c2 GEMV launches.  The old build is a copy kept out of the tree's way (tools/_ab/, not committed)."""
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
for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096)):
    out[f"gemv_n{N}k{K}"] = round(bench.time_member_gemv(dev, gen, N, K)["us_per_launch"], 3)
r = bench.time_step_chained(dev, gen)
out["step_chained_fused"] = round(r["fused"]["us_per_step"], 2)
print(json.dumps(out))
''' % ROOT


def run(lib):
    env = dict(os.environ)
    if lib:
        env["WQAA_LIBRARY"] = lib
    else:
        env.pop("WQAA_LIBRARY", None)
    a = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    members = json.loads(a.stdout.strip().splitlines()[-1]) if a.returncode == 0 else {"error": a.stderr[-400:]}
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-members", "--no-cpu-baseline", "--no-live-pmc", "--steps", "300",
                        "--warmup", "30"], env=env, capture_output=True, text=True, timeout=600)
    try:
        members["headline_us_per_step"] = round(json.loads(b.stdout.strip().splitlines()[-1])["ms_per_step"] * 1e3, 2)
    except Exception:  # noqa: BLE001
        members["headline_error"] = b.stderr[-300:]
    return members


def main():
    old = os.path.abspath(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for r in range(rounds):
        for name, lib in (("old", old), ("new", None)):
            print(json.dumps({"build": name, **run(lib)}), flush=True)


if __name__ == "__main__":
    main()
