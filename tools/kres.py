#!/usr/bin/env python
"""Per-kernel register / scratch usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
    python tools/kres.py tools/gemm_lab.hip [-I dir ...]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result",
       "-I", "bitblas_amd/csrc", "--cuda-device-only", "-S", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
for name, (k, r) in zip(names, rows.items()):
    name = re.sub(r"^void wqaa::|\(wqaa::GemmArgs\)$", "", name)
    print(f"{r.get('VGPRs', '?'):>4} vgpr {r.get('AGPRs', 0):>4} agpr {r.get('SGPRs', '?'):>4} sgpr  spill {r.get('VGPRs Spill', 0):>4}  scratch {r.get('ScratchSize', 0):>5}  {name[:110]}")
