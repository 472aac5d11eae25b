#!/usr/bin/env python
"""tools/ab_grid.py - GEMV grid size sweep on many-row matrices (M = 1, int4 g128): WQAA_GEMVX_GRID / WQAA_GEMV_GRID
(single launches) and WQAA_GROUP_GRID (workgroups per member of a group launch).  hipGraph replays, us per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    cases = [(22016, 4096, False, [688, 1024, 1376]), (28672, 4096, False, [1200, 1792, 2048, 3584]),
             (28672, 8192, False, [1200, 1792, 2048, 3584]), (57344, 8192, False, [1792, 2048, 2392, 3584, 7168]),
             (28672, 8192, True, [1200, 1792, 2048, 3584]), (22016, 4096, True, [1376, 2048, 2752])]
    for (N, K, strict, grids) in cases:
        op = bench.get_op(1, N, K, strict=strict)
        nset = max(3, (640 << 20) // (N * K // 2))
        sets = [bench.make_linear(N, K, dev, gen)[1:3] for _ in range(nset)]
        A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
        out = torch.empty((1, N), dtype=torch.float16, device=dev)
        nbytes = bench.algorithmic_bytes(1, N, K)

        def launch_all():
            st = torch.cuda.current_stream(dev).cuda_stream
            for (w, sc) in sets:
                op.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr(), None, None, out.data_ptr(), 1, st)

        var = "WQAA_GEMV_GRID" if strict else "WQAA_GEMVX_GRID"
        row = []
        for rep in range(2):
            for g in grids:
                os.environ[var] = str(g)
                plan = op.lib.plan(1)
                t = bench.graph_time(dev, launch_all, nset, replays=7)
                row.append(f"{plan['grid']}: {t * 1e6:.2f}")
        del os.environ[var]
        plan = op.lib.plan(1)
        t = bench.graph_time(dev, launch_all, nset, replays=7)
        print(f"{'strict' if strict else 'exact '} {N}x{K} [{plan['name']}] default grid {plan['grid']}: {t * 1e6:.2f} us | " + "  ".join(row))
        del sets
    # group launch: gate/up of the headline step
    Ns, K = [11008, 11008], 4096
    ops = [bench.get_op(1, N, K) for N in Ns]
    nset = 14
    sets = [[bench.make_linear(N, K, dev, gen)[1:3] for N in Ns] for _ in range(nset)]
    A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
    outs = [torch.empty((1, N), dtype=torch.float16, device=dev) for N in Ns]

    def grouped():
        for s in sets:
            bitblas.matmul_group(ops, A, s, outputs=outs)

    row = []
    for rep in range(2):
        for g in (344, 512, 688):
            os.environ["WQAA_GROUP_GRID"] = str(g)
            plan = bitblas.group_plan(ops, 1)
            t = bench.graph_time(dev, grouped, nset, replays=7)
            row.append(f"{plan['plan']['grid']}: {t * 1e6:.2f}")
    del os.environ["WQAA_GROUP_GRID"]
    print("group gate_up 2 x 11008x4096 | " + "  ".join(row))


if __name__ == "__main__":
    main()
