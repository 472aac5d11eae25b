#!/usr/bin/env python
"""Every WQAA_* environment variable the library, the Python package and bench.py read, with where and what class it is:

    python tools/list_env_knobs.py [--out profiles/rNN_env_knobs.txt]

product  = changes what a caller gets and is documented in README / INTEGRATION (opt-in vendor library, packer threads, ...)
aid      = A/B and tuning aid of tools/ and tests/: pins a member the selector would otherwise choose; never needed by a caller
A knob read in csrc/ is read at PLAN time only (selection is memoised per descriptor; wqaa_select bumps the epoch)."""
from __future__ import annotations

import argparse
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the two tuning variables hold lists of key[=value] tokens (csrc/wqaa_common.h: knob); their keys are listed under them
TUNE = {"WQAA_GEMV_TUNE": "gemv_knob", "WQAA_GEMM_TUNE": "gemm_knob"}
PRODUCT = {
    "WQAA_DENSE_LIB": "opt-in: plain dense pairs / the tuned two-pass member through the vendor library (yardstick; default off)",
    "WQAA_TWO_PASS": "opt-in: B_decode to a scratch + dense GEMM at large M (needs WQAA_DENSE_LIB for the vendor GEMM)",
    "WQAA_PACK_THREADS": "host threads of the weight packer (wqaa_pack_weight)",
    "WQAA_LIBRARY": "path of the libwqaa_hip.so to load instead of the in-tree build (two-build A/B)",
    "WQAA_PLAN_LOG": "file that receives (m, plan name) of every launch's selection (tools/member_coverage.py)",
    "WQAA_BENCH_FORCE_DIST": "bench.py: run the N > 1 code path with one rank",
    "WQAA_BENCH_GATHER": "bench.py: how the per-step all-gather is scheduled (serial / overlap / eager)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    args = ap.parse_args()
    seen = collections.defaultdict(list)
    keys = {v: collections.defaultdict(list) for v in TUNE}
    files = [os.path.join(dp, f) for top in ("bitblas_amd", ".") for dp, _, fs in os.walk(os.path.join(ROOT, top))
             for f in fs if f.endswith((".hip", ".h", ".py")) and "/tools" not in dp and "/tests" not in dp and "/oracle" not in dp
             and "/.git" not in dp and (top != "." or dp == os.path.join(ROOT, "."))]
    for path in sorted(set(files)):
        with open(path) as fh:
            for no, line in enumerate(fh, 1):
                for m in re.finditer(r'(?:getenv\(|knob\(|environ(?:\.get|\.pop|\.setdefault)?[\(\[])\s*"(WQAA_[A-Z0-9_]+)"', line):
                    seen[m.group(1)].append(f"{os.path.relpath(path, ROOT)}:{no}")
                for var, fn in TUNE.items():
                    for m in re.finditer(fn + r'(?:_set)?\("([a-z0-9_]+)"', line):
                        keys[var][m.group(1)].append(f"{os.path.relpath(path, ROOT)}:{no}")
    for var in TUNE:
        if keys[var]:
            seen.setdefault(var, ["csrc/wqaa_common.h (knob)"])
    lines = [f"# {len(seen)} WQAA_* environment variables ({sum(k in PRODUCT for k in seen)} product switches, "
             f"{sum(k not in PRODUCT for k in seen)} A/B aids); tools/list_env_knobs.py"]
    for cls in ("product", "aid"):
        for k in sorted(seen):
            if (k in PRODUCT) != (cls == "product"):
                continue
            lines.append(f"{cls:8s}{k:28s}{', '.join(seen[k])}" + (f"   - {PRODUCT[k]}" if k in PRODUCT else ""))
            for key in sorted(keys.get(k, {})):
                lines.append(f"{'':8s}  {key + '=':26s}{', '.join(keys[k][key])}")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text)
    print(text, end="")


if __name__ == "__main__":
    main()
