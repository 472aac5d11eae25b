"""Build libwqaa_hip.so (gfx950) in-tree with hipcc.  No JIT, no cmake: a handful of .hip files.

    python -m bitblas_amd.build [--force] [--jobs N]

The reference compiles one wrapper per tuned config at run time (bitblas/builder/lib_generator/
__init__.py:31-107: `hipcc -std=c++17 -fPIC --shared`); here the whole static kernel library is
compiled once, ahead of time, and travels with the source tree.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwqaa_hip.so")
OBJ_DIR = os.path.join(CSRC, "_obj")
ARCH = "gfx950"

SOURCES = ["wqaa_abi.hip", "wqaa_gemv.hip", "wqaa_gemm.hip"]
HEADERS = ["wqaa_common.h", "wqaa_decode.h", os.path.join("..", "..", "include", "wqaa.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _newest_header() -> float:
    extra = [f for f in os.listdir(CSRC) if f.endswith(".h")]
    return max(_mtime(os.path.join(CSRC, h)) for h in set(HEADERS) | set(extra))


def compile_one(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if not force and _mtime(obj) > max(_mtime(srcp), _newest_header()):
        return obj
    # --offload-compress: the device code objects travel zstd-compressed inside the library (the HIP runtime unpacks them when the
    # module loads): the static kernel library is ~4x smaller on disk and in every snapshot pushed to a GPU box
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "--offload-compress", "-c", srcp, "-o", obj,
           "-Wno-unused-result", "-ffp-contract=off"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
    return obj


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    extra = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in sources)
    sources += extra
    jobs = jobs or min(len(sources), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: compile_one(s, force), sources))
    if force or _mtime(LIB) < max(_mtime(o) for o in objs):
        # (hipBLASLt - the opt-in yardstick of csrc/wqaa_dense_lib.hip - is dlopen'ed when first asked for, never linked)
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} objects")
    return LIB


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    args = ap.parse_args(argv)
    build(force=args.force, jobs=args.jobs, verbose=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
