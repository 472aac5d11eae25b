import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bitblas_amd as bitblas
from helpers import make_case, hip_output
def run(case, env, strict):
    for k in ("WQAA_GEMV_NO_DIRECT", "WQAA_GEMVX_AREG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    got, mm = hip_output(case, strict_reference=strict)
    return got, mm.plans[case["M"]]["name"]
for (M, N, K) in ((1, 4096, 4096), (1, 1024, 4096), (2, 2048, 2048), (1, 1536, 1024)):
    for kw in (dict(W_dtype="int4", group_size=128, with_scaling=True), dict(W_dtype="uint4", group_size=128, with_scaling=True, with_zeros=True),
               dict(W_dtype="int2", group_size=64, with_scaling=True), dict(W_dtype="nf4", group_size=128, with_scaling=True)):
        case = make_case(M, N, K, seed=N + K, **kw)
        a, na = run(case, {}, True); b, nb = run(case, {"WQAA_GEMV_NO_DIRECT": "1"}, True)
        print("strict", M, N, K, kw["W_dtype"], na, "|", nb, "equal" if np.array_equal(a, b) else "DIFFER %d" % int((a != b).sum()))
        if kw["W_dtype"] in ("int4", "uint4", "int2") and M == 1:
            a, na = run(case, {"WQAA_GEMVX_AREG": "1"}, False); b, nb = run(case, {"WQAA_GEMVX_AREG": "0"}, False)
            print("exact ", M, N, K, kw["W_dtype"], na, "|", nb, "equal" if np.array_equal(a, b) else "DIFFER %d" % int((a != b).sum()))
