#!/usr/bin/env python
"""tools/ab_knobs.py "N K" ["N K" ...] -- "VAR=v VAR=v" ["VAR=v" ...]: same-process sweep of tile-selector tuning
variables (e.g. "WQAA_GEMV_TUNE=kw=2,grid=512") on M = 1 int4 g128 GEMVs (exact-product members unless AB_STRICT=1).  hipGraph replays over rotating
weight sets, two rounds, microseconds per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    shapes = [tuple(int(x) for x in a.split()) for a in argv[:cut]]
    combos = [""] + argv[cut + 1:]
    strict = os.environ.get("AB_STRICT", "0") == "1"
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    for (N, K) in shapes:
        op = bench.get_op(1, N, K, strict=strict)
        nset = max(3, min(64, (640 << 20) // (N * K // 2)))
        sets = [bench.make_linear(N, K, dev, gen)[1:3] for _ in range(nset)]
        A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
        out = torch.empty((1, N), dtype=torch.float16, device=dev)

        def launch_all():
            st = torch.cuda.current_stream(dev).cuda_stream
            for (w, sc) in sets:
                op.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr(), None, None, out.data_ptr(), 1, st)

        res = {}
        for rnd in range(2):
            for combo in combos:
                kv = dict(x.split("=", 1) for x in combo.split()) if combo else {}
                os.environ.update(kv)
                try:
                    plan = op.lib.plan(1)
                    t = bench.graph_time(dev, launch_all, nset, replays=7)
                    res.setdefault(combo, [plan["name"].split("_", 2)[2] + f" g{plan['grid']} t{plan['threads']}"]).append(t * 1e6)
                except Exception as e:  # noqa: BLE001
                    res.setdefault(combo, [f"refused: {e}"])
                for k in kv:
                    del os.environ[k]
        op.lib.plan(1)
        for combo, v in res.items():
            print(f"{N}x{K} {combo or 'default':44s} {v[0]:44s} " + "  ".join(f"{x:6.2f}" for x in v[1:]))
        del sets


if __name__ == "__main__":
    main()
