import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, bitblas_amd as bitblas
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for w in ("int4", "int2"):
    for M in (4096, 2048):
        N = K = 4096
        op = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int4", W_dtype=w, accum_dtype="int32", out_dtype="int32"), enable_tuning=False)
        A = torch.randint(-128, 128, (M, K // 2), device=dev, dtype=torch.int8, generator=gen)
        Ws = [torch.randint(-128, 128, (N, K * op.bit // 8), dtype=torch.int8, device=dev, generator=gen) for _ in range(4)]
        out = torch.empty((M, N), dtype=torch.int32, device=dev)
        def launch_all():
            for W in Ws:
                op(A, W, output=out)
        t = bench.graph_time(dev, launch_all, 4)
        print(json.dumps({"w": w, "M": M, "plan": op.plans[M]["name"], "us": round(t * 1e6, 1), "TOPS": round(2.0 * M * N * K / t / 1e12, 1)}), flush=True)
