#!/usr/bin/env python
"""tools/ab_group_knobs.py -- "VAR=v ..." ...: tile-selector tuning variables on the two group launches of the headline
step (q/k/v 3 x 4096x4096, gate/up 2 x 11008x4096, M = 1 int4 g128), same process, two rounds, us per group launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402


def main():
    argv = sys.argv[1:]
    combos = [""] + (argv[argv.index("--") + 1:] if "--" in argv else [])
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (name, Ns, K) in (("qkv", [4096] * 3, 4096), ("gate_up", [11008] * 2, 4096)):
        total = sum(Ns)
        nset = max(4, (640 << 20) // (total * K // 2))
        A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
        sets = [[bench.make_linear(N, K, dev, gen)[1:3] for N in Ns] for _ in range(nset)]
        ops = [bench.get_op(1, N, K) for N in Ns]
        outs = [torch.empty((1, N), dtype=torch.float16, device=dev) for N in Ns]

        def grouped():
            for s in sets:
                bitblas.matmul_group(ops, A, s, outputs=outs)

        res = {}
        for rnd in range(2):
            for combo in combos:
                kv = dict(x.split("=") for x in combo.split()) if combo else {}
                os.environ.update(kv)
                plan = bitblas.group_plan(ops, 1)
                t = bench.graph_time(dev, grouped, nset, replays=7)
                p = plan["plan"]
                res.setdefault(combo, [(p["name"].split("_", 2)[2] + f" g{p['grid']} t{p['threads']}") if p else "unfused"]).append(t * 1e6)
                for k in kv:
                    del os.environ[k]
        bitblas.group_plan(ops, 1)
        for combo, v in res.items():
            print(f"{name:8s} {combo or 'default':44s} {v[0]:48s} " + "  ".join(f"{x:6.2f}" for x in v[1:]))


if __name__ == "__main__":
    main()
