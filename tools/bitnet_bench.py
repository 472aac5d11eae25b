"""BitNet-style linear, M=1..16, N=K=4096: fused boundary ops (HIP quantiser + matmul_ex epilogue) vs the
reference's structure (torch quantiser -> int32/fp32-out matmul -> torch `out / si / sw -> half`), both
captured in a hipGraph.  Prints microseconds per layer call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitblas_amd as bitblas
from bitblas_amd.bitnet import BitLinear


def graph_us(fn, reps=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(ts))


def main():
    N = K = 4096
    w = torch.randn(N, K, device="cuda") * 0.02
    lin = BitLinear(K, N).cuda()
    lin.load_float_weight(w)
    mm = bitblas.Matmul(bitblas.MatmulConfig(M=[1, 16], N=N, K=K, A_dtype="int8", W_dtype="int2", out_dtype="float32",
                                             accum_dtype="int32"), enable_tuning=False)
    sw = lin.sw
    for m in (1, 16):
        x = torch.randn(m, K, device="cuda", dtype=torch.float16)

        def unfused():
            xf = x.float()
            s = 127.0 / xf.abs().max(dim=-1, keepdim=True).values.clamp(min=1e-5)
            q = (xf * s).round().clamp(-128, 127).to(torch.int8)
            out = mm(q, lin.qweight)
            return ((out / s) / sw).half()

        a, b = lin(x), unfused()
        assert torch.allclose(a.float(), b.float(), rtol=2e-3, atol=2e-3), (a - b).abs().max()  # torch GPU division is not IEEE-exact
        print(f"M={m}: fused {graph_us(lambda: lin(x)):.2f} us/call   reference structure {graph_us(unfused):.2f} us/call")

    # q/k/v of a block: three layers one by one against BitLinearGroup (wqaa_matmul_group_ex: one launch at m <= 2)
    from bitblas_amd.bitnet import BitLinearGroup
    qkv = []
    for _ in range(3):
        l = BitLinear(K, N).cuda()
        l.load_float_weight(torch.randn(N, K, device="cuda") * 0.02)
        qkv.append(l)
    grp = BitLinearGroup(qkv)
    for m in (1, 2):
        x = torch.randn(m, K, device="cuda", dtype=torch.float16)
        for a, b in zip([l(x) for l in qkv], grp(x)):
            assert torch.equal(a, b)
        print(f"M={m} q/k/v (3 x {N}x{K}): layers one by one {graph_us(lambda: [l(x) for l in qkv]):.2f} us   "
              f"BitLinearGroup {graph_us(lambda: grp(x)):.2f} us")


if __name__ == "__main__":
    main()
