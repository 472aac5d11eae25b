#!/usr/bin/env python
"""Same-process sweep of the gate/up pair launch's tile knobs (workgroup width, grid cap) against the group launch, hipGraph replays
over rotating weight sets: python tools/ab_pair_knobs.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402
from bitblas_amd import lib as wlib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    H, I, n = 4096, 11008, 24
    gates = [bench.make_linear(I, H, dev, gen) for _ in range(n)]
    ups = [bench.make_linear(I, H, dev, gen) for _ in range(n)]
    x = (torch.rand((1, H), device=dev, generator=gen) - 0.5).to(torch.float16)
    act = torch.empty((1, I), dtype=torch.float16, device=dev)

    def group_only():
        for g, u in zip(gates, ups):
            bitblas.matmul_group([g[0], u[0]], x, [(g[1], g[2]), (u[1], u[2])], outputs=[g[3], u[3]])

    def pair():
        for g, u in zip(gates, ups):
            bitblas.matmul_gate_up(g[0], u[0], x, (g[1], g[2]), (u[1], u[2]), output=act)

    def t(fn):
        return bench.graph_time(dev, fn, n) * 1e6

    print(f"group launch alone            {t(group_only):7.2f} {t(group_only):7.2f}")
    for env in ({}, {"WQAA_GEMVX_SLOTS": "4"}, {"WQAA_GEMVX_SLOTS": "16"}, {"WQAA_GEMVX_GRID": "1024"}, {"WQAA_GEMVX_GRID": "2048", "WQAA_GEMVX_SLOTS": "4"},
                {"WQAA_GEMVX_GRID": "2752", "WQAA_GEMVX_SLOTS": "4"}, {"WQAA_GEMV_UNCAP": "1"}):
        for k in ("WQAA_GEMVX_SLOTS", "WQAA_GEMVX_GRID", "WQAA_GEMV_UNCAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        wlib.select(gates[0][0].lib.desc, 1)          # bumps the plan epoch: the next launch re-reads the environment
        plan = bitblas.gate_up_plan(gates[0][0], 1)
        print(f"pair {str(env):60s} threads {plan['threads']:4d} grid {plan['grid']:5d}  {t(pair):7.2f} {t(pair):7.2f}")


if __name__ == "__main__":
    main()
