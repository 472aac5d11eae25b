#!/usr/bin/env python
"""tools/ab_bf16.py: W_uint4 x A_bfloat16 (g128 + zeros, bfloat16 output) GEMM - the selector's tile against the lockstep member
(WQAA_GEMM_PP_BM=0), same process, hipGraph replays over rotating weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bitblas_amd as bitblas
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (M, N, K) in ((4096, 4096, 4096), (2048, 4096, 4096), (4096, 11008, 4096)):
    row = []
    for rnd in range(2):
        for force in (None, "0"):
            if force is None: os.environ.pop("WQAA_GEMM_PP_BM", None)
            else: os.environ["WQAA_GEMM_PP_BM"] = force
            op = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="bfloat16", W_dtype="uint4", out_dtype="bfloat16", accum_dtype="float32",
                                                     group_size=128, with_scaling=True, with_zeros=True), enable_tuning=False)
            A = (torch.rand((M, K), device=dev, generator=gen) - 0.5).bfloat16()
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            sets = [(torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=gen),
                     (torch.rand((N, K // 128), device=dev, generator=gen) * 0.02).bfloat16(),
                     torch.full((N, K // 128), 8.0, dtype=torch.bfloat16, device=dev)) for _ in range(6)]
            def launch_all():
                st = torch.cuda.current_stream(dev).cuda_stream
                for qw, sc, zr in sets:
                    op.lib.run(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), zr.data_ptr(), None, out.data_ptr(), M, st)
            t = bench.graph_time(dev, launch_all, len(sets))
            row.append((op.plans[M]["name"].split("_")[-1], t * 1e6, 2.0 * M * N * K / t / 1e12))
    os.environ.pop("WQAA_GEMM_PP_BM", None)
    print(f"bf16 x uint4 M={M} N={N} K={K}: " + "  ".join(f"{n} {u:7.1f} us ({tf:5.0f} TF)" for n, u, tf in row))
