"""End-to-end `Matmul.forward` / `Linear.forward` rate in eager mode (Python + ctypes + launch), next to the
kernel-only time of the same call replayed from a hipGraph.  SURVEY 8(d): report both."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bitblas_amd as bitblas

def main():
    dev = torch.device("cuda")
    for M in (1, 16):
        cfg = bitblas.MatmulConfig(M=M, N=4096, K=4096, A_dtype="float16", W_dtype="int4", out_dtype="float16",
                                   accum_dtype="float16", group_size=128, with_scaling=True)
        mm = bitblas.Matmul(cfg, enable_tuning=False)
        A = (torch.rand(M, 4096, device=dev) - 0.5).half()
        W = torch.randint(-128, 127, (4096, 2048), device=dev, dtype=torch.int8)
        S = torch.rand(4096, 32, device=dev).half() * 0.02
        out = torch.empty(M, 4096, device=dev, dtype=torch.float16)
        for _ in range(200):
            mm(A, W, scale=S, output=out)
        torch.cuda.synchronize()
        n = 5000
        t0 = time.perf_counter()
        for _ in range(n):
            mm(A, W, scale=S, output=out)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                mm(A, W, scale=S, output=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 5000 * 1e6
        lin = bitblas.Linear(4096, 4096, bias=False, A_dtype="float16", W_dtype="int4", accum_dtype="float16", out_dtype="float16",
                             group_size=128, with_scaling=True, opt_M=[1, 16], enable_tuning=False).to(dev)
        for _ in range(200):
            lin(A)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            lin(A)
        torch.cuda.synchronize()
        lin_eager = (time.perf_counter() - t0) / n * 1e6
        print(f"M={M}: Matmul.forward eager {eager:.2f} us/call, hipGraph replay {graph:.2f} us/call, Linear.forward eager {lin_eager:.2f} us/call")

def groups():
    """q/k/v as three eager `Linear.forward` calls, as one `LinearGroup` call, and as `matmul_group` with preallocated outputs"""
    dev = torch.device("cuda")
    layers = [bitblas.Linear(4096, 4096, bias=False, A_dtype="float16", W_dtype="int4", accum_dtype="float16", out_dtype="float16",
                             group_size=128, with_scaling=True, opt_M=[1, 16], enable_tuning=False).to(dev) for _ in range(3)]
    grp = bitblas.LinearGroup(layers).to(dev)
    A = (torch.rand(1, 4096, device=dev) - 0.5).half()
    ops = [l.bitblas_matmul for l in layers]
    ws = [(l.qweight, l.scales) for l in layers]
    outs = [torch.empty(1, 4096, device=dev, dtype=torch.float16) for _ in range(3)]
    res = {}
    for name, fn in (("3 x Linear.forward", lambda: [l(A) for l in layers]), ("LinearGroup.forward", lambda: grp(A)),
                     ("matmul_group(outputs=...)", lambda: bitblas.matmul_group(ops, A, ws, outputs=outs))):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n = 3000
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n * 1e6
    print("q/k/v 3 x 4096^2 int4 g128, M=1, eager, us per group: " + ", ".join(f"{k} {v:.2f}" for k, v in res.items()))


if __name__ == "__main__":
    main()
    groups()
