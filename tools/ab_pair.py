#!/usr/bin/env python
"""Same-process A/B of the fused elementwise ops: hipGraph replays over rotating weight sets, us per launch sequence.
  gate/up:  group launch alone | group + torch silu, mul | wqaa_matmul_gate_up (one launch, stores the activation only)
  o / down: plain launch | plain + torch add | WQAA_EPI_ADD_RESIDUAL
Usage: python tools/ab_pair.py [--sets 24]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    H, I = 4096, 11008
    n = args.sets
    gates = [bench.make_linear(I, H, dev, gen) for _ in range(n)]
    ups = [bench.make_linear(I, H, dev, gen) for _ in range(n)]
    downs = [bench.make_linear(H, I, dev, gen) for _ in range(n)]
    os_ = [bench.make_linear(H, H, dev, gen) for _ in range(n)]
    x = (torch.rand((1, H), device=dev, generator=gen) - 0.5).to(torch.float16)
    a = (torch.rand((1, I), device=dev, generator=gen) - 0.5).to(torch.float16)
    act = torch.empty((1, I), dtype=torch.float16, device=dev)
    res = torch.zeros((1, H), dtype=torch.float16, device=dev)
    out = torch.empty((1, H), dtype=torch.float16, device=dev)

    def t(fn):
        return bench.graph_time(dev, fn, n) * 1e6

    def group_only():
        for g, u in zip(gates, ups):
            bitblas.matmul_group([g[0], u[0]], x, [(g[1], g[2]), (u[1], u[2])], outputs=[g[3], u[3]])

    def group_torch():
        for g, u in zip(gates, ups):
            bitblas.matmul_group([g[0], u[0]], x, [(g[1], g[2]), (u[1], u[2])], outputs=[g[3], u[3]])
            torch.mul(torch.nn.functional.silu(g[3]), u[3], out=act)

    def pair():
        for g, u in zip(gates, ups):
            bitblas.matmul_gate_up(g[0], u[0], x, (g[1], g[2]), (u[1], u[2]), output=act)

    rows = [("gate/up 2 x 11008 x 4096: group launch alone", t(group_only)), ("  group + torch silu, mul", t(group_torch)),
            ("  wqaa_matmul_gate_up", t(pair))]
    for name, lins, inp in (("down 4096 x 11008", downs, a), ("o 4096 x 4096", os_, x)):
        def plain():
            for d in lins:
                d[0].forward(inp, d[1], scale=d[2], output=out)

        def plain_add():
            for d in lins:
                d[0].forward(inp, d[1], scale=d[2], output=out)
                out.add_(res)

        def fused():
            for d in lins:
                d[0].forward_ex(inp, d[1], scale=d[2], residual=res, output=out)
        rows += [(f"{name}: plain launch", t(plain)), ("  plain + torch add", t(plain_add)), ("  WQAA_EPI_ADD_RESIDUAL", t(fused))]
    for k, v in rows:
        print(f"{k:52s} {v:8.2f} us")


if __name__ == "__main__":
    main()
