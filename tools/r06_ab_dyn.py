#!/usr/bin/env python
"""Round 6: run-time hand-out of row groups (members `..._dyn`, WQAA_GEMV_TUNE=dyn=1) against the static members, same process:
bit-identity first, then hipGraph replays over rotating weights (bench.py's harness), two alternating rounds.  Single GEMVs and the
gate / up group launches of a 7B and a 70B layer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bitblas_amd as bitblas  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1)


def arm(env):
    os.environ.pop("WQAA_GEMV_TUNE", None)
    if env:
        os.environ["WQAA_GEMV_TUNE"] = env


for (N, K) in ((28672, 8192), (8192, 28672), (8192, 8192), (10240, 8192), (22016, 4096), (12288, 4096), (14336, 4096), (32000, 4096)):
    op = bench.get_op(1, N, K)
    nset = max(3, min(64, (640 << 20) // (N * K // 2)))
    sets = [bench.make_linear(N, K, dev, gen)[1:3] for _ in range(nset)]
    A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
    outs = {}
    names = {}
    for tag, env in (("static", "dyn=0"), ("dyn", "dyn=1")):
        arm(env)
        names[tag] = op.lib.plan(1)["name"].split("_", 2)[2]
        o = torch.empty((1, N), dtype=torch.float16, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(3):      # repeated launches: the hand-out words must come back to zero
            o.zero_()
            op.lib.run(A.data_ptr(), sets[0][0].data_ptr(), None, sets[0][1].data_ptr(), None, None, o.data_ptr(), 1, st)
        torch.cuda.synchronize()
        outs[tag] = o
    same = bool(torch.equal(outs["static"], outs["dyn"]))
    out = torch.empty((1, N), dtype=torch.float16, device=dev)

    def launch_all():
        st = torch.cuda.current_stream(dev).cuda_stream
        for (w, sc) in sets:
            op.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr(), None, None, out.data_ptr(), 1, st)

    res = {"static": [], "dyn": []}
    for rnd in range(2):
        for tag, env in (("static", "dyn=0"), ("dyn", "dyn=1")):
            arm(env)
            op.lib.plan(1)
            res[tag].append(bench.graph_time(dev, launch_all, nset, replays=7) * 1e6)
    arm(None)
    op.lib.plan(1)
    print(f"{N}x{K}  bit-identical {same}   static {names['static']:28s} " + " ".join(f"{x:6.2f}" for x in res["static"]) +
          f"   dyn {names['dyn']:32s} " + " ".join(f"{x:6.2f}" for x in res["dyn"]), flush=True)
    del sets

# group launches: gate / up of a Llama-2-7B layer (2 x 11008 x 4096) and of a 70B layer (2 x 28672 x 8192)
for (Ns, K) in (((11008, 11008), 4096), ((4096, 4096, 4096), 4096), ((28672, 28672), 8192)):
    ops = [bench.get_op(1, N, K) for N in Ns]
    wbytes = sum(N * K // 2 for N in Ns)
    nset = max(3, min(32, (640 << 20) // wbytes))
    sets = [[bench.make_linear(N, K, dev, gen)[1:3] for N in Ns] for _ in range(nset)]
    A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
    got = {}
    for tag, env in (("static", "dyn=0"), ("dyn", "dyn=1")):
        arm(env)
        bitblas.group_plan(ops, 1)
        o = bitblas.matmul_group(ops, A, sets[0])
        o = bitblas.matmul_group(ops, A, sets[0])
        torch.cuda.synchronize()
        got[tag] = [x.clone() for x in o]
    same = all(torch.equal(a, b) for a, b in zip(got["static"], got["dyn"]))
    outs = [torch.empty((1, N), dtype=torch.float16, device=dev) for N in Ns]

    def launch_all():
        for ws in sets:
            bitblas.matmul_group(ops, A, ws, outputs=outs)

    res = {"static": [], "dyn": []}
    name = {}
    for rnd in range(2):
        for tag, env in (("static", "dyn=0"), ("dyn", "dyn=1")):
            arm(env)
            name[tag] = (bitblas.group_plan(ops, 1)["plan"] or {}).get("name", "?").split("_", 2)[2]
            res[tag].append(bench.graph_time(dev, launch_all, nset, replays=7) * 1e6)
    arm(None)
    bitblas.group_plan(ops, 1)
    print(f"group {Ns} K={K}  bit-identical {same}   static {name['static']} " + " ".join(f"{x:6.2f}" for x in res["static"]) +
          f"   dyn {name['dyn']} " + " ".join(f"{x:6.2f}" for x in res["dyn"]), flush=True)
    del sets
