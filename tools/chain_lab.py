"""Lab for wqaa_matmul_chain (csrc/wqaa_chain_kernel.h): the post-attention half of a Llama-2-7B decoder layer at M = 1,
o_proj (+ x) -> RMSNorm -> gate / up * silu -> down_proj (+ h), as ONE persistent launch against the three launches it stands
for (forward_ex, matmul_gate_up, forward_ex), hipGraph replays over rotating weight sets; bit-compared; with the in-kernel
time line of one traced launch (s_memrealtime stamps per wave: loader first DMA / stream done, per stage stager start / tile
ready / tasks done).
    python tools/chain_lab.py [--layers 6] [--reps 5]      # one JSON line per variant + the time line table"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bitblas_amd as bitblas
from bitblas_amd.chain import ChainStep, chain_plan, chain_status, chain_trace, matmul_chain

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--hidden", type=int, default=4096)
ap.add_argument("--inter", type=int, default=11008)
ap.add_argument("--no-trace", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
H, I, G = args.hidden, args.inter, 128


def op(N, K):
    return bitblas.Matmul(bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype="int4", out_dtype="float16", accum_dtype="float16",
                                               group_size=G, with_scaling=True), enable_tuning=False)


def lin(o):
    return (torch.randint(-128, 128, (o.N, o.K // 2), dtype=torch.int8, device=dev, generator=gen),
            (torch.rand((o.N, o.K // G), device=dev, generator=gen) * 0.02 * 0.04).to(torch.float16))


o_op, g_op, d_op = op(H, H), op(I, H), op(H, I)
layers = [dict(o=lin(o_op), g=lin(g_op), u=lin(g_op), d=lin(d_op),
               nw=(1.0 + (torch.rand(H, device=dev, generator=gen) - 0.5) * 0.2).to(torch.float16)) for _ in range(args.layers)]
attn = (torch.rand((1, H), device=dev, generator=gen) - 0.5).to(torch.float16)
x0 = (torch.rand((1, H), device=dev, generator=gen) - 0.5).to(torch.float16)
eps = 1e-5
hs = [torch.empty((1, H), dtype=torch.float16, device=dev) for _ in range(args.layers)]
acts = [torch.empty((1, I), dtype=torch.float16, device=dev) for _ in range(args.layers)]
outs = [torch.empty((1, H), dtype=torch.float16, device=dev) for _ in range(args.layers)]


def run_launches():
    x = x0
    for L, h, a, o in zip(layers, hs, acts, outs):
        o_op.forward_ex(attn, L["o"][0], scale=L["o"][1], residual=x, output=h)
        bitblas.matmul_gate_up(g_op, g_op, h, L["g"], L["u"], output=a, norm=(L["nw"], eps))
        d_op.forward_ex(a, L["d"][0], scale=L["d"][1], residual=h, output=o)
        x = o
    return x


def steps_of(L, x, h=None, a=None, o=None):
    return [ChainStep(o_op, L["o"], attn, residual=x, output=h if h is not None else False),
            ChainStep(g_op, L["g"], 0, norm=(L["nw"], eps), up_op=g_op, up_weights=L["u"], output=a if a is not None else False),
            ChainStep(d_op, L["d"], 1, residual=0, output=o)]


chs, cas, cos = [torch.empty_like(t) for t in hs], [torch.empty_like(t) for t in acts], [torch.empty_like(t) for t in outs]


def run_chain(keep):
    x = x0
    for i, L in enumerate(layers):
        matmul_chain(steps_of(L, x, chs[i] if keep else None, cas[i] if keep else None, cos[i]))
        x = cos[i]
    return x


def graph_us(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    per = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        per.append(e0.elapsed_time(e1) * 1e3 / args.layers)
    return per


def variant(name, env):
    """one configuration of the persistent member (plan-time switches: chain_plan re-reads them)"""
    for k in ("WQAA_CHAIN_LANES", "WQAA_CHAIN_CPL", "WQAA_CHAIN_THIN", "WQAA_CHAIN_SWEEP_SLEEP", "WQAA_CHAIN_RING", "WQAA_CHAIN_TRACE", "WQAA_CHAIN_LAB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    plan = chain_plan(steps_of(layers[0], x0, chs[0], cas[0], cos[0]))
    for t in chs + cas + cos:
        t.zero_()
    run_chain(True); torch.cuda.synchronize()
    st = chain_status()
    same = [bool(torch.equal(a, b)) for a, b in zip(hs + acts + outs, chs + cas + cos)]
    per = graph_us(lambda: run_chain(False))
    rec = {"variant": name, "env": env, "plan": (plan["plan"] or {}).get("name"), "threads": (plan["plan"] or {}).get("threads"),
           "error": st["error"], "bit_identical_stages": f"{sum(same)}/{len(same)}",
           "us_per_layer_tail": [round(p, 2) for p in per], "median": round(float(np.median(per)), 2),
           "GBps": round(nbytes / np.median(per) / 1e3, 1)}
    print(json.dumps(rec), flush=True)
    return rec


def timeline(name, env):
    for k in ("WQAA_CHAIN_LANES", "WQAA_CHAIN_CPL", "WQAA_CHAIN_THIN", "WQAA_CHAIN_SWEEP_SLEEP", "WQAA_CHAIN_RING", "WQAA_CHAIN_TRACE", "WQAA_CHAIN_LAB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["WQAA_CHAIN_TRACE"] = "1"
    plan = chain_plan(steps_of(layers[0], x0, None, None, cos[0]))
    matmul_chain(steps_of(layers[0], x0, None, None, cos[0]))
    torch.cuda.synchronize()
    matmul_chain(steps_of(layers[1], x0, None, None, cos[1]))       # cold weights for the traced launch
    torch.cuda.synchronize()
    tr = chain_trace().astype(np.int64)
    os.environ.pop("WQAA_CHAIN_TRACE")
    for k in env:
        os.environ.pop(k, None)
    chain_plan(steps_of(layers[0], x0, None, None, cos[0]))
    if not tr.size:
        return
    nl = 4 if not env.get("WQAA_CHAIN_LANES") else int(env["WQAA_CHAIN_LANES"])
    t0 = tr[:, :, 0][tr[:, :, 0] > 0].min()
    names = {0: "wave start", 1: "loader: first DMA issued", 2: "loader: stream drained", 3: "wave end", 20: "loader 0: 16 units issued",
             21: "loader 0: 32 units issued", 22: "loader 0: 48 units issued", 23: "loader 0: 64 units issued"}
    for s in range(3):
        names[4 + 3 * s] = f"stage {s}: staging starts"
        names[5 + 3 * s] = f"stage {s}: input tile ready"
        names[6 + 3 * s] = f"stage {s}: this wave's tasks done"
    ok = (tr[:, :, 24] > 0) & (tr[:, :, 25] > tr[:, :, 24]) & (tr[:, :, 3] > tr[:, :, 0])
    mhz = 2350.0
    if ok.any():
        m = (tr[:, :, 25] - tr[:, :, 24])[ok] / ((tr[:, :, 3] - tr[:, :, 0])[ok] / 100.0)
        mhz = float(np.median(m))
        print(f"[{name}] {(plan['plan'] or {}).get('name')}: shader clock over the launch: min {m.min():.0f} median {mhz:.0f} max {m.max():.0f} MHz")
    print(f"time line of one launch [{name}], microseconds after the first wave's start (100 MHz clock): min / median / max over the waves that stamped")
    for i in sorted(names):
        v = tr[:, :, i]
        v = v[v > t0 - 100000]
        if v.size:
            u = (v - t0) / 100.0
            print(f"  {names[i]:44s} n={v.size:4d}  {u.min():7.2f} {np.median(u):7.2f} {u.max():7.2f}")
    cons = tr[:, nl:, :]
    live = cons[:, :, 3] > 0
    print("  shader-clock marks per consumer wave, microseconds since the wave's stage-loop entry of the stage (median over waves; '-' = not passed):")
    mk = ("9 stage entered", "0 tile free", "1 my passes gathered", "2 meeting A passed", "3 my ladder done", "4 meeting B passed", "5 scaled, before last meeting",
          "6 after last meeting", "7 task context ready", "8 tasks done")
    order = (9, 0, 1, 2, 3, 4, 5, 6, 7, 8)
    for st in range(3):
        base = cons[:, :, 32 + 10 * st + 9].astype(np.float64)
        row = []
        for i in order:
            v = cons[:, :, 32 + 10 * st + i].astype(np.float64)
            okm = (v > 0) & (base > 0)
            row.append(f"{np.median((v - base)[okm]) / mhz:6.2f}" if okm.any() else "     -")
        print(f"    stage {st}: " + "  ".join(f"{n.split(' ')[0]}:{r}" for n, r in zip([mk[i if i != 9 else 0] if False else str(i) for i in order], row)))
    print("    (marks: 9 stage entered, 0 tile free, 1 my passes gathered, 2 meeting A passed, 3 my ladder done, 4 meeting B passed, 5 before the last meeting, 6 after it, 7 task context ready, 8 tasks done)")
    print("  per consumer wave, microseconds of shader time (min / median / max): ")
    for i, nm in enumerate(("waiting for weights to land", "in the tasks (dots, store, publish)", "staging inputs (all of it)", "... of which in the meetings",
                            "... of which waiting for granules")):
        v = cons[:, :, 26 + i][live] / mhz
        print(f"    {nm:40s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")


run_launches(); torch.cuda.synchronize()
nbytes = sum(2 * t.numel() if t.dtype == torch.float16 else t.numel() for L in layers[:1] for k in "ogud" for t in L[k])
per = graph_us(run_launches)
print(json.dumps({"variant": "launches_3_per_layer", "us_per_layer_tail": [round(p, 2) for p in per], "median": round(float(np.median(per)), 2),
                  "GBps": round(nbytes / np.median(per) / 1e3, 1), "weight_bytes_per_tail": nbytes}), flush=True)
# lab bits (results wrong by construction): 1 no dots, 2 consumers do not wait for the weights, 4 no sweeps, 8 default-policy DMA
VARIANTS = [("l4c2 (default)", {}), ("l4c3", {"WQAA_CHAIN_CPL": "3"}), ("l4c1", {"WQAA_CHAIN_CPL": "1"}), ("l2c2", {"WQAA_CHAIN_LANES": "2"}),
            ("l4c2_nothin", {"WQAA_CHAIN_THIN": "0"}),
            ("lab1_no_dots", {"WQAA_CHAIN_LAB": "1"}), ("lab4_no_sweeps", {"WQAA_CHAIN_LAB": "4"}), ("lab5_no_dots_no_sweeps", {"WQAA_CHAIN_LAB": "5"}),
            ("lab21_no_stream_no_sweeps_no_dots", {"WQAA_CHAIN_LAB": "21"})]
for name, env in VARIANTS:
    variant(name, env)
per = graph_us(run_launches)
print(json.dumps({"variant": "launches_again", "median": round(float(np.median(per)), 2)}), flush=True)
if not args.no_trace:
    timeline("l4c2 (default)", {})
    timeline("lab21_no_stream_no_sweeps_no_dots", {"WQAA_CHAIN_LAB": "21"})
