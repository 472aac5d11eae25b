#!/usr/bin/env python
"""Static resource report of every kernel in the library, from the compiler's own metadata - no GPU needed.

    python tools/static_isa_report.py [--out profiles/rNN_static_isa.txt] [--jobs 8]

Compiles each csrc/*.hip device-only to assembly (`hipcc --cuda-device-only -S`, same flags as bitblas_amd/build.py),
reads the `amdhsa.kernels` metadata (.vgpr_count, .sgpr_count, .private_segment_fixed_size = scratch bytes per lane,
.group_segment_fixed_size = static LDS, .max_flat_workgroup_size) and reports

  * every kernel that uses scratch (register spills / private arrays) - and, for each, whether a scratch access sits
    inside a loop of its body (the expensive kind);
  * waves per SIMD the register files admit: VGPR (512 per lane per SIMD, granule 8, unified arch + acc) and SGPR
    (800 per SIMD, granule 16 + 16 reserved: <= 80 -> 8 waves, 82-96 -> 7, 98-112 -> 6; MI355X_MICROARCH.md, occupancy notes);
  * a per-family summary.

It is a lint, not a measurement: what it finds are candidates for a same-box A/B on the GPU."""
from __future__ import annotations

import argparse
import collections
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bitblas_amd", "csrc")
HOST_ONLY = {"wqaa_pack.hip"}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise SystemExit("hipcc not found")


def to_asm(src, outdir):
    out = os.path.join(outdir, os.path.basename(src).replace(".hip", ".s"))
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result",
           "--cuda-device-only", "-S", src, "-o", out]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{src}: {res.stderr[-2000:]}")
    return out


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [o.replace("wqaa::", "") for o in out[:len(names)]]
    except FileNotFoundError:
        return list(names)


def waves_by_vgpr(v):
    g = max(8, (v + 7) // 8 * 8)
    return min(8, 512 // g)


def waves_by_sgpr(s):
    g = (s + 15) // 16 * 16 + 16
    return min(8, 800 // g)


def scratch_in_loop(body):
    """does a scratch access sit between a loop header label and the backward branch to it?"""
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    hits = [i for i, l in enumerate(lines) if "scratch_" in l and "scratch_en" not in l]
    inside = sum(1 for h in hits if any(a <= h <= b for a, b in loops))
    return len(hits), inside


def parse(asm):
    txt = open(asm).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size:", txt, re.S):
        blk = m.group(0)

        def g(k):
            mm = re.search(r"\.%s:\s+(\S+)" % k, blk)
            return mm.group(1) if mm else "0"
        name = g("name")
        row = dict(file=os.path.basename(asm)[:-2], name=name, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")),
                   sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                   lds=int(g("group_segment_fixed_size")), wg=int(g("max_flat_workgroup_size")), n_scr=0, n_scr_loop=0)
        if row["scratch"]:
            s = txt.find("\n" + name + ":")
            e = txt.find(".Lfunc_end", s)
            if s >= 0 and e > s:
                row["n_scr"], row["n_scr_loop"] = scratch_in_loop(txt[s:e])
        rows.append(row)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 4))
    ap.add_argument("--asm-dir", default=None, help="reuse / keep the generated .s files here")
    args = ap.parse_args()
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and f not in HOST_ONLY)
    outdir = args.asm_dir or tempfile.mkdtemp(prefix="wqaa_isa_")
    os.makedirs(outdir, exist_ok=True)
    todo = [s for s in srcs if not os.path.exists(os.path.join(outdir, os.path.basename(s).replace(".hip", ".s")))]
    with cf.ThreadPoolExecutor(max_workers=args.jobs) as pool:
        list(pool.map(lambda s: to_asm(s, outdir), todo))
    rows = []
    for s in srcs:
        rows += parse(os.path.join(outdir, os.path.basename(s).replace(".hip", ".s")))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["pretty"] = re.sub(r"\((Gem\w+Args|.*)\)$", "", d)[:118]
    lines = []
    w = lines.append
    w(f"static ISA report: {len(rows)} kernels in {len(srcs)} translation units (hipcc -O3 --offload-arch=gfx950; compiler metadata)")
    w("")
    fam = collections.defaultdict(list)
    for r in rows:
        fam[r["file"]].append(r)
    w(f"{'translation unit':30s} {'kernels':>7s} {'vgpr min/med/max':>18s} {'sgpr max':>8s} {'with scratch':>12s} {'sgpr>80':>8s}")
    for f, rs in sorted(fam.items()):
        v = sorted(r["vgpr"] for r in rs)
        w(f"{f:30s} {len(rs):7d} {v[0]:6d}/{v[len(v) // 2]:4d}/{v[-1]:4d}   {max(r['sgpr'] for r in rs):8d} "
          f"{sum(1 for r in rs if r['scratch']):12d} {sum(1 for r in rs if r['sgpr'] > 80):8d}")
    w("")
    spill = [r for r in rows if r["scratch"]]
    w(f"kernels with scratch: {len(spill)}  (scratch = bytes per lane; 'in loop' = scratch instructions between a loop header and its back edge)")
    for r in sorted(spill, key=lambda r: (-r["n_scr_loop"], -r["scratch"])):
        w(f"  {r['file']:26s} {r['pretty']:118s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:5d} B  "
          f"instr {r['n_scr']:3d} (in loop {r['n_scr_loop']:3d})  wg<={r['wg']}")
    w("")
    w("GEMV-family kernels whose SGPR count, not their VGPR count, limits the waves per SIMD (the actionable kind: scalars are cheap to shed):")
    n = 0
    for r in rows:
        wv, ws = waves_by_vgpr(r["vgpr"] + r["agpr"]), waves_by_sgpr(r["sgpr"])
        if ws < wv and "gemm_" not in r["file"]:
            n += 1
            w(f"  {r['file']:26s} {r['pretty']:118s} vgpr {r['vgpr']:3d} -> {wv}  sgpr {r['sgpr']:3d} -> {ws}")
    w(f"  ({n} kernels)")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
