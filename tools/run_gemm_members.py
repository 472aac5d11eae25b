"""the MFMA members bench.py reports, under rocprofv3 --kernel-trace (tools/r05_final.sh): M = 4096 fused uint4 / int2 x int8 (+ BitNet
epilogue), the mid-M member of BASELINE c3's M = 128 (its two kernels), M = 16, decode batches on long K (the K-sliced form) and wide N, dense float16 and the e4m3 Llama-3-70B linears (c5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, torch, bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
keys = ("kernel", "us_per_launch", "TFLOPs", "frac_of_mfma_peak")
for kw in (dict(), dict(bitnet=True)):
    r = bench.time_member_gemm(dev, gen, 4096, W_dtype="int2", A_dtype="int8", **kw)
    print(json.dumps({k: r.get(k) for k in keys}))
for M in (4096, 128, 96, 16):
    r = bench.time_member_gemm(dev, gen, M)
    print(json.dumps({k: r.get(k) for k in keys}))
r = bench.time_member_gemm(dev, gen, 64, 4096, 8192)
print(json.dumps({k: r.get(k) for k in keys}))
for (M, N, K) in ((8, 8192, 28672), (16, 8192, 28672), (8, 22016, 4096)):      # decode batches: the K-sliced form (its two kernels), the persistent form
    r = bench.time_member_gemm(dev, gen, M, N, K)
    print(json.dumps({k: r.get(k) for k in keys}))
r = bench.time_member_dense(dev, gen, 4096, 4096, 4096, kind="f16", n_buf=4)
print(json.dumps({k: r.get(k) for k in keys}))
for (N, K, nb) in ((8192, 8192, 4), (8192, 28672, 2), (10240, 8192, 4), (28672, 8192, 2)):
    r = bench.time_member_dense(dev, gen, 4096, N, K, n_buf=nb)
    print(json.dumps({k: r.get(k) for k in keys}))
