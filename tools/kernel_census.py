#!/usr/bin/env python
"""tools/kernel_census.py [kernel_stats.csv ...]: every kernel of the built library (the code objects' metadata) against the kernels a
profiled run launched (rocprofv3 --kernel-trace --stats of the GPU parity suite: `*_kernel_stats.csv`).  Prints how many kernels the
library holds, how many the run launched, and - grouped by template - the ones no test ever launched."""
import collections
import csv
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def library_kernels(path):
    spec = importlib.util.spec_from_file_location("check_vmem_hazards", os.path.join(ROOT, "tools", "check_vmem_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    names = set()
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(mod.code_objects(path)):
            f = os.path.join(tmp, f"co{k}.elf")
            with open(f, "wb") as fh:
                fh.write(co)
            notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
            names.update(re.findall(r"\.name:\s+(\S+)", notes))
    return {n for n in names if not n.endswith(".kd")}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def norm(s):
    s = re.sub(r"\s+", "", s)
    s = re.sub(r"^void", "", s)
    s = re.sub(r"\[clone\.kd\]$", "", s)
    return re.sub(r"\(.*$", "", s)          # without the argument list (kernel-trace names stop in front of it, or not: both forms)


def main():
    lib = os.path.join(ROOT, "bitblas_amd", "libwqaa_hip.so")
    kernels = sorted(library_kernels(lib))
    dm = demangle(kernels)
    launched = set()
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                launched.add(norm(row.get("Name") or row.get("Kernel_Name") or ""))
    by_tpl = collections.Counter()
    dead = collections.defaultdict(list)
    n_hit = 0
    for k in kernels:
        d = norm(dm[k])
        tpl = d.split("<")[0]
        by_tpl[tpl] += 1
        if d in launched:
            n_hit += 1
        else:
            dead[tpl].append(dm[k])
    print(f"{len(kernels)} kernels in {os.path.relpath(lib, ROOT)}; {n_hit} launched by the profiled run(s); {len(kernels) - n_hit} never launched")
    for tpl, n in by_tpl.most_common():
        print(f"  {tpl:48s} {n:5d} instantiated  {n - len(dead.get(tpl, [])):5d} launched")
    if "--list" in sys.argv:
        for tpl in dead:
            for d in sorted(dead[tpl]):
                print("DEAD", d)


if __name__ == "__main__":
    main()
