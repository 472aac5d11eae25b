import torch, numpy as np, sys
sys.path.insert(0, '.')
import bitblas_amd as bitblas
for M in (1, 3, 16, 64, 300):
    for (N, K) in ((256, 512), (256, 256)):
        mm = bitblas.Matmul(bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float32"), enable_tuning=False)
        A = torch.full((M, K), 0x38, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)   # 1.0
        Wb = torch.zeros((N, K), dtype=torch.uint8, device="cuda")
        Wb[:, :] = 0x03      # subnormal 3/8 * 2^-6
        W = Wb.view(torch.float8_e4m3fn)
        out = mm(A, W)
        torch.cuda.synchronize()
        exp = K * (3/8) * 2**-6
        # and subnormal activations against normal weights
        A2 = torch.full((M, K), 0x05, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)
        W2 = torch.full((N, K), 0x40, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)   # 2.0
        out2 = mm(A2, W2); torch.cuda.synchronize()
        exp2 = K * (5/8) * 2**-6 * 2.0
        print(M, N, K, mm.plans[M]["name"], "W-subnormal:", out[0,0].item(), "expected", exp, "| A-subnormal:", out2[0,0].item(), "expected", exp2)
