#!/usr/bin/env python
"""Round 4, lab (NOT in the tree any more: profiles/r04_ps_check.txt): the post-scale member of the one-launch decode member (WQAA_GEMM_DECODE_POSTSCALE=1: group scale applied to the fp32
partial sums after the MFMA, csrc/wqaa_gemm_kernel.h FL_PS) - achieved error against the CPU oracle next to the default member's,
and time per launch (hipGraph replays) against the default member."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from helpers import hip_output, make_case, oracle_output


def err(got, want):
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    rms = max(float(np.sqrt(np.mean(want ** 2))), 1e-30)
    e = np.abs(got - want)
    big = np.abs(want) > 0.1 * rms
    return {"max_rel": round(float((e[big] / np.abs(want[big])).max()), 6), "max_abs_over_rms": round(float(e.max() / rms), 6), "finite": bool(np.isfinite(got).all())}


cases = [dict(M=8, N=11008, K=4096, W_dtype="uint4", with_zeros=True, zeros_mode="original"),
         dict(M=16, N=11008, K=4096, W_dtype="uint4", with_zeros=True, zeros_mode="original"),
         dict(M=3, N=22016, K=4096, W_dtype="uint4", with_zeros=True, zeros_mode="original"),
         dict(M=5, N=8192 + 48, K=1024, W_dtype="uint4", with_zeros=True, zeros_mode="rescale", with_bias=True),
         dict(M=12, N=4096 + 64, K=1024, W_dtype="int4"),
         dict(M=8, N=8192, K=8192, W_dtype="uint4", with_zeros=True, zeros_mode="original"),
         dict(M=4, N=8192, K=11008, W_dtype="uint4", with_zeros=True, zeros_mode="original"),
         dict(M=4, N=11008, K=3840, W_dtype="uint4", with_zeros=True, zeros_mode="original")]
for c in cases:
    M, N, K = c.pop("M"), c.pop("N"), c.pop("K")
    case = make_case(M, N, K, group_size=128, with_scaling=True, scale_mul=0.05, seed=M + N, **c)
    want = oracle_output(case)
    os.environ.pop("WQAA_GEMM_DECODE_POSTSCALE", None)
    got0, mm0 = hip_output(case)
    os.environ["WQAA_GEMM_DECODE_POSTSCALE"] = "1"
    got1, mm1 = hip_output(case)
    os.environ.pop("WQAA_GEMM_DECODE_POSTSCALE", None)
    print(json.dumps({"M": M, "N": N, "K": K, **{k: str(v) for k, v in c.items()}, "default": (mm0.plans[M]["name"].rsplit("_", 1)[1], err(got0, want)),
                      "postscale": (mm1.plans[M]["name"].rsplit("_", 1)[1], err(got1, want))}), flush=True)

dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (M, N, K) in ((8, 11008, 4096), (16, 11008, 4096), (3, 11008, 4096), (8, 22016, 4096), (8, 8192, 8192), (8, 12288, 8192)):
    row = {"M": M, "N": N, "K": K}
    for name, env in (("default", {}), ("postscale", {"WQAA_GEMM_DECODE_POSTSCALE": "1"})):
        os.environ.pop("WQAA_GEMM_DECODE_POSTSCALE", None); os.environ.update(env)
        bench._OPS.clear()
        r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="uint4")
        row[name] = (r["kernel"].rsplit("_", 1)[1], round(r["us_per_launch"], 2))
    os.environ.pop("WQAA_GEMM_DECODE_POSTSCALE", None)
    print(json.dumps(row), flush=True)
