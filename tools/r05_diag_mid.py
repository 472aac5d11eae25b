import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import numpy as np
from helpers import make_case, hip_output, oracle_output
for seam in ("0", "1"):
    os.environ["WQAA_GEMM_MID_SEAM"] = seam
    for M in (100, 128, 97, 112, 65, 100):
        case = make_case(M, 4096, 4096, W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True, zeros_mode="original", scale_mul=0.02, seed=M)
        got, mm = hip_output(case)
        want = oracle_output(case)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = err > 1e-3 * np.abs(want) + 1e-3 * np.sqrt(np.mean(want.astype(np.float64) ** 2))
        rows = np.where(bad.any(axis=1))[0]
        cols = np.where(bad.any(axis=0))[0]
        print(f"seam={seam} M={M} {mm.plans[M]['name']} bad={int(bad.sum())} rows={rows[:20].tolist()} ncols={len(cols)} cols[:8]={cols[:8].tolist()}", flush=True)
