#!/usr/bin/env python
"""tools/ab_direct_fit.py: same-process A/B of the rounding GEMV members whose register-resident ("areg") form spills
(profiles/r02_static_isa.txt): default selector vs WQAA_GEMV_DIRECT_FIT=1 (LDS-staged twin
wherever the activation slice does not fit the register file).  hipGraph replays over rotating weight sets, two rounds,
microseconds per launch; results of both members are compared bit for bit first."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402

CASES = [("uint1", "float16", 1), ("uint1", "float16", 2), ("uint2", "float16", 2), ("uint2", "float16", 1),
         ("int1", "int8", 2), ("uint4", "float16", 2)]
SHAPES = [(4096, 4096), (11008, 4096), (1024, 8192)]


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (wd, ad, M) in CASES:
        bits = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[wd][1]
        int8 = ad == "int8"
        for (N, K) in SHAPES:
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=ad, W_dtype=wd, out_dtype="int32" if int8 else "float16",
                                       accum_dtype="int32" if int8 else "float16", group_size=-1 if int8 else 128,
                                       with_scaling=not int8)
            op = bitblas.Matmul(cfg, enable_tuning=False)
            nset = max(3, min(64, (640 << 20) // (N * K * bits // 8)))
            sets = [(torch.randint(-128, 128, (N, K * bits // 8), dtype=torch.int8, device=dev, generator=gen),
                     None if int8 else (torch.rand((N, K // 128), device=dev, generator=gen) * 0.02).half()) for _ in range(nset)]
            A = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=gen) if int8 else \
                (torch.rand((M, K), device=dev, generator=gen) - 0.5).half()
            out = torch.empty((M, N), dtype=torch.int32 if int8 else torch.float16, device=dev)

            def launch_all():
                st = torch.cuda.current_stream(dev).cuda_stream
                for (w, sc) in sets:
                    op.lib.run(A.data_ptr(), w.data_ptr(), None, None if sc is None else sc.data_ptr(), None, None, out.data_ptr(), M, st)

            res, outs = {}, {}
            for rnd in range(2):
                for combo in ("", "WQAA_GEMV_DIRECT_FIT=1"):
                    if combo:
                        os.environ["WQAA_GEMV_DIRECT_FIT"] = "1"
                    plan = op.lib.plan(M)
                    t = bench.graph_time(dev, launch_all, nset, replays=7)
                    outs[combo] = out.clone()
                    res.setdefault(combo, [plan["name"].split("_", 2)[2]]).append(t * 1e6)
                    os.environ.pop("WQAA_GEMV_DIRECT_FIT", None)
            op.lib.plan(M)
            same = torch.equal(outs[""], outs["WQAA_GEMV_DIRECT_FIT=1"])
            for combo, v in res.items():
                print(f"{wd:6s} {ad:8s} M={M} {N}x{K} {combo or 'default':26s} {v[0]:34s} " + "  ".join(f"{x:6.2f}" for x in v[1:]) +
                      ("" if same else "   RESULTS DIFFER"))
            del sets


if __name__ == "__main__":
    main()
