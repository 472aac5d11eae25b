import json, os, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
for (M, N, K) in ((4096, 10240, 8192), (2048, 10240, 8192), (4096, 3584, 8192), (4096, 1280, 8192), (4096, 28672, 8192), (3072, 10240, 8192)):
    row = {}
    for rep in range(2):
        for name, env in (("default", {}), ("no_tail", {"WQAA_GEMM_PP_TAIL": "0"})):
            os.environ.pop("WQAA_GEMM_PP_TAIL", None); os.environ.update(env)
            bench._OPS.clear()
            r = bench.time_member_dense(dev, gen, M, N, K, kind="fp8", n_buf=2)
            row.setdefault(name, []).append((r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 1)))
    os.environ.pop("WQAA_GEMM_PP_TAIL", None)
    print(json.dumps({"shape": [M, N, K], **row}), flush=True)
# int2 x int8 partial rounds
for (M, N, K) in ((2048, 11008, 4096), (4096, 11008, 4096)):
    row = {}
    for name, env in (("default", {}), ("no_tail", {"WQAA_GEMM_PP_TAIL": "0"})):
        os.environ.pop("WQAA_GEMM_PP_TAIL", None); os.environ.update(env)
        bench._OPS.clear()
        r = bench.time_member_gemm(dev, gen, M, N, K, W_dtype="int2", A_dtype="int8")
        row[name] = (r["kernel"].split("_", 2)[2], round(r["us_per_launch"], 1))
    os.environ.pop("WQAA_GEMM_PP_TAIL", None)
    print(json.dumps({"i2 shape": [M, N, K], **row}), flush=True)
