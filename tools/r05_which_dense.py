import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitblas_amd as bitblas
M = N = K = 1024
A = (torch.rand((M, K), device="cuda") - 0.5).half()
W = (torch.rand((N, K), device="cuda") - 0.5).half()
mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="float16", accum_dtype="float32", out_dtype="float16"), enable_tuning=False)
print(mm.plans[M]["name"])
for _ in range(3):
    mm(A, W)
torch.cuda.synchronize()
