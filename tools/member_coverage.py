#!/usr/bin/env python
"""tools/member_coverage.py [plan.log ...]: which kernel members did a parity run exercise?

    WQAA_PLAN_LOG=gpurun_out/plan.log python -m pytest tests -m gpu -q        # on the GPU box
    python tools/member_coverage.py gpurun_out/plan.log                       # anywhere

`WQAA_PLAN_LOG` makes the Python layer append the plan name of every (operator, row count) it launches (bitblas_amd/lib.py).
A plan name minus its shape - `f16xu4_tcx64x128x128xrxw`, `i8xi2_gemv_b1r1d2_areg`, `f16xi4_gemvx_b1r2d2k1_x3` - names a member
CLASS: activation x weight types, family, tile / variant suffix.  This tool prints the classes in the log and, from a
device-less sweep of the selector over the operator's configuration space (dtype pairs x zero modes x layouts x row counts x
shapes that hit every rule of the selectors), the classes the selector can reach that the log does not contain.  Without a
log it prints the reachable set only."""
import itertools
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bitblas_amd as bitblas  # noqa: E402


def member_class(name: str) -> str:
    m = re.match(r"^matmul_m\d+n\d+k\d+_(.*)$", name)
    cls = m.group(1) if m else name
    return re.sub(r"_x\d+$", "_xG", cls)            # group launches: one class whatever the member count


def mode_tag(mode, fd):
    """the template arguments a plan name does not show: scale / zeros mode and checkpoint layout"""
    z = {"original": "zo", "rescale": "zr", "quantized": "zq"}[mode["zeros_mode"]] if mode.get("with_zeros") else ("s" if mode.get("with_scaling") else "none")
    return z + ("" if fd is None else "_plain")


def reachable(with_args=False, per_kernel=False):
    """class -> a one-line example (or, with_args, the example's arguments: tests/test_member_coverage_gpu.py runs it).
    per_kernel: one entry per (class, scale / zeros mode, layout) - about one per kernel INSTANTIATION a plain call can reach
    (round 6: a rocprofv3 census of the GPU suite found two thirds of the library's kernels launched by no test,
    tools/kernel_census.py); examples come from the smallest shape that reaches the entry"""
    f16 = [("float16", w) for w in ("uint4", "int4", "uint2", "int2", "uint1", "int1", "uint8", "int8", "nf4", "fp4_e2m1", "e4m3_float8", "float16")]
    bf16 = [("bfloat16", w) for w in ("uint4", "int4", "uint2", "uint1", "int8", "nf4", "fp4_e2m1", "e4m3_float8", "bfloat16")]
    if per_kernel:
        bf16.append(("bfloat16", "uint8"))
    i8 = [("int8", w) for w in ("int8", "int4", "uint4", "int2", "uint2", "int1")]
    f8 = [("e4m3_float8", "e4m3_float8"), ("e5m2_float8", "e5m2_float8")]
    if per_kernel:              # (the mixed pairs of general_matmul/__init__.py:33-51, and e5m2 weights under float16 activations: :344)
        f8 += [("e4m3_float8", "e5m2_float8"), ("e5m2_float8", "e4m3_float8")]
        f16 = f16 + [("float16", "e5m2_float8")]
    i4 = [("int4", "int4"), ("int4", "int2")]
    shapes = [(1024, 1024), (4096, 4096), (11008, 4096), (4096, 11008), (1024, 28672), (28672, 8192), (272, 2048), (5120, 4096), (2048, 8192), (4352, 8192)]
    if per_kernel:
        # (N off the MFMA family's multiple of 4, K off its k grid: the GEMV family's 4-row batch tile; few rows x long K: its K-split twins)
        shapes = sorted([(64, 256), (48, 128), (100, 384), (272, 512), (528, 1024), (2048, 1024), (8192, 512), (16384, 256), (1024, 8192), (512, 16384), (24, 4096),
                         (50, 256), (50, 2048), (64, 160), (64, 192), (24, 32768), (1024, 320),
                         (50, 4096), (50, 8192), (50, 16384), (2048, 16384), (4096, 384),
                         (32768, 128),               # (a round of 256 x 256 tiles at 512 rows: the lockstep tile of the formats whose 1024 rows go B_decode + dense)
                         (4092, 768)] + shapes,     # (4092: N off the ping-pong members' multiple of 8 at M = 4096 - the 256-row lockstep tile of every format)
                        key=lambda nk: nk[0] * nk[1])
    ms = [1, 2, 3, 8, 16, 32, 64, 128, 256, 1024, 4096] if not per_kernel else [1, 2, 3, 5, 8, 16, 32, 64, 128, 256, 512, 1024, 4096]
    seen = {}
    for (a, w) in f16 + bf16 + i8 + f8 + i4:
        quant = a in ("float16", "bfloat16") and w not in (a, "fp4_e2m1")
        modes = [dict()]
        if quant:
            modes.append(dict(with_scaling=True, group_size=128))
            if w.startswith("uint") and (w != "uint8" or per_kernel):
                modes += [dict(with_scaling=True, group_size=128, with_zeros=True, zeros_mode=z) for z in ("original", "rescale", "quantized")]
            if per_kernel:
                modes += [dict(with_scaling=True, group_size=32)] + ([dict(with_scaling=True, group_size=32, with_zeros=True, zeros_mode="original")] if w.startswith("uint") else [])
                if w == "uint2":            # (K = 192: off the MFMA k grid, groups of 64 = one lane chunk - the 4-row GEMV tile with packed zero points)
                    modes.append(dict(with_scaling=True, group_size=64, with_zeros=True, zeros_mode="quantized"))
        fds = [None, False] if (w[0] in "ui" and w not in ("uint8", "int8") and a in ("float16", "int8")) else [None]
        if per_kernel and a == "int8" and w in ("int4", "uint4"):
            fds = [None, True]          # (the reference's default for this pair is the plain layout: general_matmul/__init__.py:171-173)
        if per_kernel and (a, w) == ("int4", "int2"):
            fds = [None, False]
        for (N, K), mode, fd, strict in itertools.product(shapes, modes, fds, (True, False)):
            # strict_reference picks other members for sub-byte integers x float16 (per-element rounding) and for e4m3 x float16 (the reference's bit trick)
            strict_matters = a == "float16" and ((w[0] in "ui" and w not in ("uint8", "int8")) or (per_kernel and w == "e4m3_float8"))
            if not strict and not strict_matters:
                continue
            out = "int32" if a in ("int8", "int4") else ("bfloat16" if a == "bfloat16" else "float16")
            acc = "int32" if a in ("int8", "int4") else "float32"
            cfg = dict(M=ms, N=N, K=K, A_dtype=a, W_dtype=w, out_dtype=out, accum_dtype=acc, fast_decoding=fd, **mode)
            try:
                op = bitblas.Matmul(bitblas.MatmulConfig(**cfg), enable_tuning=False, strict_reference=strict)
            except Exception:  # noqa: BLE001 - a refused configuration reaches no member
                continue
            for m in ms:
                key = member_class(op.plans[m]["name"])
                if per_kernel:
                    key += "|" + mode_tag(mode, fd) + (f"_g{mode['group_size']}" if mode.get("group_size", 128) != 128 else "") + ("_fd" if fd else "")
                    key += "|default" if (strict_matters and not strict and w == "e4m3_float8") else ""
                if with_args:
                    seen.setdefault(key, dict(M=m, N=N, K=K, a=a, w=w, mode=mode, fd=fd, strict=strict, cfg=cfg))
                else:
                    seen.setdefault(key, f"M={m} N={N} K={K} {a} x {w} {mode} fd={fd} strict={strict}")
    return seen


def main():
    logs = [a for a in sys.argv[1:] if os.path.exists(a)]
    reach = reachable()
    print(f"{len(reach)} member classes reachable by the selector over the sweep (no device: own members for the plain dense pairs)")
    if not logs:
        for c in sorted(reach):
            print("  ", c)
        return
    hit = {}
    for path in logs:
        for line in open(path):
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 2:
                hit[member_class(parts[1].split("+")[0])] = hit.get(member_class(parts[1].split("+")[0]), 0) + 1
    print(f"{len(hit)} member classes in the log(s) ({sum(hit.values())} (operator, row count) pairs)")
    missing = sorted(c for c in reach if c not in hit)
    print(f"{len(missing)} reachable classes NOT exercised by the logged run:")
    for c in missing:
        print(f"   {c:44s} e.g. {reach[c]}")
    extra = sorted(c for c in hit if c not in reach)
    print(f"{len(extra)} logged classes outside the sweep (vendor-library members, forced variants): " + ", ".join(extra[:40]))


if __name__ == "__main__":
    main()
