#!/usr/bin/env python
"""bench.py - the headline measurement (BASELINE.json): W_int4 A_fp16 GEMV at M=1 on the
Llama-2-7B linear shapes (N, K in {4096, 11008}), group_size=128, on MI355X - BASELINE `configs[1]`.
One JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one decode pass over LAYERS synthetic decoder layers: each layer = the 7 linear GEMVs of a
Llama-2-7B block (q,k,v,o: 4096x4096; gate,up: 11008x4096; down: 4096x11008), every layer with its
own weight buffers, so a step streams LAYERS x 105 MB of distinct packed weights (> the 256 MiB
Infinity Cache): the bytes really come from HBM.  The projections of a layer that read the same input -
q/k/v and gate/up - go through `bitblas_amd.matmul_group` (wqaa_matmul_group: one launch per group, every
member with its own packed tensors and output; the reference's integration fuses the same projections by
concatenating their weights, integration/BitNet/modeling_bitnet.py:1433-1445), so a layer is 4 launches:
{q,k,v}, o, {gate,up}, down.  `--no-groups` launches the 7 GEMVs one by one (also timed under
`members["step_ungrouped"]`).  The launches of a step are captured once into a
hipGraph (the GEMV is ~4-8 us: eager Python launches would measure the interpreter) and a step is
one graph replay.  Inputs are resident in HBM before the timed region.
value = algorithmic bytes moved per second by the whole job (GB/s).

roofline: every launch of the step is the same kernel function (the int4 GEMV family member); its
average launch duration is measured live with events on the launch stream around a replayed graph
of those launches: `achieved` = average algorithmic bytes per launch / average duration.  The
N=K=4096 member named in BASELINE.json's target and the M=4096 MFMA GEMM (TFLOP/s vs the dense
fp16 MFMA peak) are timed the same way and reported under `members`.

N > 1: the weight matrices are column(N)-sharded: every rank owns an equal slice of every layer
(weak scaling: per-GPU work fixed, i.e. the model is N_gpus x wider) and the per-layer output slices
are all-gathered with one RCCL all-gather per step, captured at the end of the step's hipGraph (one replay per
step, no per-step host call into RCCL; WQAA_BENCH_GATHER=eager|overlap select the measured alternatives, DESIGN.md
section 6).  WQAA_BENCH_FORCE_DIST=1 runs that code path with a single rank (1-GPU boxes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bitblas_amd as bitblas  # noqa: E402

LLAMA2_7B_LINEARS = [  # (name, N, K)
    ("q_proj", 4096, 4096), ("k_proj", 4096, 4096), ("v_proj", 4096, 4096), ("o_proj", 4096, 4096),
    ("gate_proj", 11008, 4096), ("up_proj", 11008, 4096), ("down_proj", 4096, 11008),
]
GROUP = 128
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured on a copy)
MFMA_F16_PEAK_TF = 2500.0   # dense fp16/bf16 MFMA peak
MFMA_I8_PEAK_TOPS = 5000.0


def algorithmic_bytes(M, N, K, bits=4, g=GROUP, scale=True, zeros=False, out_bytes=2, a_bytes=2):
    """M*K*sA + N*K*bit/8 + N*(K/g)*2 [scale] + N*(K/g)*2 [zeros] + M*N*sOut"""
    b = M * K * a_bytes + N * K * bits // 8 + M * N * out_bytes
    if scale:
        b += N * (K // g) * 2
    if zeros:
        b += N * (K // g) * 2
    return b


_OPS = {}


def get_op(M, N, K, W_dtype="int4", A_dtype="float16", out_dtype="float16", zeros=False, scaling=True,
           accum="float16", strict=False):
    """strict=False: `Matmul(..., strict_reference=False)` - the exact-product members where they exist (the
    dequantised weight is not rounded to float16 per element; csrc/wqaa_gemvx_kernel.h): at least as close to the
    real-valued product as the reference's definition and inside its 1e-3 contract (tests/test_gemvx_gpu.py).
    strict=True: the TE definition's per-element rounding, bit-faithful B_decode (tests/test_te_golden.py)."""
    key = (M, N, K, W_dtype, A_dtype, out_dtype, zeros, scaling, accum, strict)
    if key not in _OPS:
        cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype=A_dtype, W_dtype=W_dtype, out_dtype=out_dtype,
                                   accum_dtype=accum, group_size=GROUP if scaling else -1, with_scaling=scaling,
                                   with_zeros=zeros)
        # strict=False: constructed exactly as a caller of the reference would (no extra argument): the library's default
        _OPS[key] = bitblas.Matmul(cfg, enable_tuning=False, strict_reference=True) if strict else bitblas.Matmul(cfg, enable_tuning=False)
    return _OPS[key]


def make_linear(N, K, device, gen):
    """One synthetic quantised linear: operator + resident operands (random codes, rand*0.02 scale)."""
    op = get_op(1, N, K)
    qweight = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=device, generator=gen)
    scale = (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)
    out = torch.empty((1, N), dtype=torch.float16, device=device)
    return op, qweight, scale, out


# which members the headline is timed on: the library's default at M <= 2, as a caller of the reference
# constructs the operator; the contract is include/wqaa.h's (at `strict_reference`), tests/helpers.py: contract
NUMERICS = "default (strict_reference=0): exact products, 1e-3 rel + 1e-3 rms vs the TE definition at K>=4096 group-wise (2e-3 rms below); *_strict: per-element rounding"

GRAPH_WARM_MS = 25.0


def graph_time(device, launch_all, n_launches, replays=5, warm_ms=None):
    """Capture `launch_all` (a sequence of kernel launches on the current stream) into one hipGraph,
    replay it, return the average duration of one launch in seconds (median over replays), measured
    with events on the stream the graph runs on.  Untimed replays first, `warm_ms` of them (default 25 ms):
    a member is timed at the clocks the chip settles at under its load, not on the ramp out of the idle state the
    host-side set-up left it in (same-box A/B of the M = 4096 GEMM: 117.7 us timed cold against 105.3 sustained,
    profiles/r04_ab_bench_warm.txt)."""
    launch_all()
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    first_ms = max(e0.elapsed_time(e1), 1e-3)
    for _ in range(min(2000, int((GRAPH_WARM_MS if warm_ms is None else warm_ms) / first_ms))):
        g.replay()
    torch.cuda.synchronize(device)
    per = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        per.append(e0.elapsed_time(e1) * 1e-3 / n_launches)
    return float(np.median(per))


def time_member_gemv(device, gen, N, K, n_buf=64, strict=False):
    """Average launch duration of the M=1 int4 GEMV at (N, K), rotating over n_buf weight sets."""
    n_buf = max(8, min(n_buf, (640 << 20) // (N * K // 2)))
    op = get_op(1, N, K, strict=strict)
    bufs = [make_linear(N, K, device, gen)[1:3] for _ in range(n_buf)]
    A = (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)
    out = torch.empty((1, N), dtype=torch.float16, device=device)

    def launch_all():
        stream = torch.cuda.current_stream(device).cuda_stream
        for qw, sc in bufs:
            op.lib.run(A.data_ptr(), qw.data_ptr(), None, sc.data_ptr(), None, None, out.data_ptr(), 1, stream)

    t = graph_time(device, launch_all, n_buf)
    nbytes = algorithmic_bytes(1, N, K)
    return {"workload": f"W_int4 A_fp16 GEMV M=1 N={N} K={K} g=128", "kernel": op.plans[1]["name"],
            "numerics": "per-element float16 rounding of B_decode (TE definition)" if strict else "exact products, group scale on fp32 partial sums",
            "us_per_launch": t * 1e6, "bytes_per_launch": nbytes, "GBps": nbytes / t / 1e9,
            "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / t / 1e9 / HBM_PEAK_GBS},
            "frac_of_hbm_peak": nbytes / t / 1e9 / HBM_PEAK_GBS, "buffers": n_buf}


def time_member_gemm(device, gen, M=4096, N=4096, K=4096, W_dtype="uint4", A_dtype="float16", n_buf=8, tuned=False, bitnet=False, frac_zeros=False):
    """MFMA GEMM members (BASELINE configs c3 / c4): TFLOP/s from graph-replayed launches.
    bitnet: the int8 member as a BitNet layer calls it (integration/BitNet/utils_quant.py:205-216) - float16 output through the fused
    `out / si / sw` epilogue (wqaa_matmul_ex) instead of the int32 sums."""
    int8 = A_dtype == "int8"
    try:
        op = get_op(M, N, K, W_dtype=W_dtype, A_dtype=A_dtype, out_dtype="float16" if (bitnet or not int8) else "int32",
                    zeros=not int8, scaling=not int8, accum="int32" if int8 else "float16")
        if op.plans[M]["kernel_family"] != 2:
            return None
        if tuned:
            # YARDSTICK, not the product: with the vendor library opted in (WQAA_DENSE_LIB=1) `Matmul.hardware_aware_finetune`
            # times the fused MFMA member against the two-pass member (B_decode to a scratch + hipBLASLt's GEMM) and keeps the faster
            os.environ["WQAA_DENSE_LIB"] = "1"
            op = bitblas.Matmul(op.config, enable_tuning=False)
            op.hardware_aware_finetune()
    except Exception as exc:  # member not built: report, never fake
        return {"error": str(exc)}
    bits = op.bit
    if M <= 256:
        # decode batches are microseconds per launch: as many launches per replayed graph as the M = 1 members take (the
        # fixed cost of a replay was a tenth of an 8-launch graph of these), over > 256 MB of distinct weights
        n_buf = max(n_buf, min(64, (640 << 20) // max(1, N * K * bits // 8)))
    if int8:
        A = torch.randint(-128, 128, (M, K), device=device, dtype=torch.int8, generator=gen)
    else:
        A = (torch.rand((M, K), device=device, generator=gen) - 0.5).to(torch.float16)
    out = torch.empty((M, N), dtype=torch.int32 if (int8 and not bitnet) else torch.float16, device=device)
    row_scale = (torch.rand((M,), device=device, generator=gen) * 50.0 + 50.0).to(torch.float32) if bitnet else None
    sets = []
    for _ in range(n_buf):
        qw = torch.randint(-128, 128, (N, K * bits // 8), dtype=torch.int8, device=device, generator=gen)
        sc = (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)
        zr = torch.full((N, K // GROUP), float(1 << (bits - 1)), dtype=torch.float16, device=device)
        if frac_zeros:
            # zero points with a fraction (the reference's `original` mode takes any float16): the ping-pong member then decodes by the
            # general (w - z) * s form instead of the integer-zero-point one GPTQ-style checkpoints get (csrc/wqaa_gemm_pp_kernel.h: zint)
            zr += 0.3125
        sets.append((qw, sc, zr))

    def launch_all():
        stream = torch.cuda.current_stream(device).cuda_stream
        for qw, sc, zr in sets:
            if bitnet:
                op.lib.run_fused(A.data_ptr(), qw.data_ptr(), None, out.data_ptr(), M, stream, row_scale.data_ptr(), 37.5)
            else:
                op.lib.run(A.data_ptr(), qw.data_ptr(), None, None if int8 else sc.data_ptr(),
                           None if int8 else zr.data_ptr(), None, out.data_ptr(), M, stream)

    try:
        t = graph_time(device, launch_all, n_buf)
    finally:
        if tuned:
            os.environ.pop("WQAA_DENSE_LIB", None)
    peak = MFMA_I8_PEAK_TOPS if int8 else MFMA_F16_PEAK_TF
    tf = 2.0 * M * N * K / t / 1e12
    nbytes = algorithmic_bytes(M, N, K, bits=bits, zeros=not int8, scale=not int8, out_bytes=4 if (int8 and not bitnet) else 2, a_bytes=1 if int8 else 2)
    # the roof that binds this shape: decode batches sit left of the ridge (4-bit weights: 2 M N K flops over ~N K / 2
    # bytes = 4 M flop/B against 2500 / 8 = 312), where the weight stream, not the matrix pipe, sets the floor
    t_mfma, t_hbm = 2.0 * M * N * K / (peak * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    if t_hbm > t_mfma:
        roof = {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                "bytes_per_launch": nbytes, "mfma_frac": tf / peak}
    else:
        roof = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s" if not int8 else "TOP/s", "frac": tf / peak,
                "flops_per_launch": 2.0 * M * N * K}
    return {"workload": f"W_{W_dtype} A_{A_dtype} GEMM M={M} N={N} K={K}" + ("" if int8 else " g=128 zeros=original"),
            "kernel": op.lib.plan_ex(M, 0)["name"] if bitnet else op.plans[M]["name"], **({"tuning": getattr(op, "_tuned", {}).get(M, "fused member kept")} if tuned else {}),
            "us_per_launch": t * 1e6, "TFLOPs": tf, "roofline": roof,
            "GBps_algorithmic": nbytes / t / 1e9, "frac_of_mfma_peak": tf / peak, "mfma_peak": peak}


def time_step_int2_int8(device, gen, n_layers=4, grouped=True):
    """BASELINE c4 at M = 1 as a STEP like c2's: a BitNet-b1.58 decoder layer's seven projections on the Llama-2-7B shapes,
    W_int2 x A_int8 -> int32 (bit exact), the projections that share an input as one launch ({q,k,v}, o, {gate,up}, down)."""
    import bitblas_amd as bitblas
    ops = {}
    layers = []
    for _ in range(n_layers):
        layer = []
        for (_, N, K) in LLAMA2_7B_LINEARS:
            if (N, K) not in ops:
                ops[(N, K)] = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="int8", W_dtype="int2", accum_dtype="int32",
                                                                  out_dtype="int32"), enable_tuning=False)
            W = torch.randint(-128, 128, (N, K // 4), dtype=torch.int8, device=device, generator=gen)
            layer.append((ops[(N, K)], W, torch.empty((1, N), dtype=torch.int32, device=device)))
        layers.append(layer)
    acts = {K: torch.randint(-128, 128, (1, K), dtype=torch.int8, device=device, generator=gen) for K in (4096, 11008)}
    groups = [[0, 1, 2], [3], [4, 5], [6]] if grouped else [[i] for i in range(7)]

    def launch_all():
        stream = torch.cuda.current_stream(device).cuda_stream
        for layer in layers:
            for grp in groups:
                if len(grp) == 1:
                    op, W, out = layer[grp[0]]
                    op.lib.run(acts[op.K].data_ptr(), W.data_ptr(), None, None, None, None, out.data_ptr(), 1, stream)
                else:
                    gops = [layer[i][0] for i in grp]
                    bitblas.matmul_group(gops, acts[gops[0].K], [layer[i][1] for i in grp], outputs=[layer[i][2] for i in grp])

    t = graph_time(device, launch_all, 1)
    nbytes = n_layers * sum(N * K // 4 + K + 4 * N for (_, N, K) in LLAMA2_7B_LINEARS)
    from bitblas_amd import group_plan
    names = {"+".join(LLAMA2_7B_LINEARS[i][0] for i in grp):
             (group_plan([layers[0][i][0] for i in grp], 1)["plan"] or layers[0][grp[0]][0].plans[1])["name"] for grp in groups}
    return {"workload": f"W_int2 A_int8 GEMV M=1 (bit exact), Llama-2-7B linear shapes, {n_layers} layers x 7 GEMV in "
                        f"{len(groups)} launches per layer, one hipGraph replay; weights rotate over {nbytes >> 20} MB",
            "launches": names, "us_per_step": t * 1e6, "bytes_per_step": nbytes, "GBps": nbytes / t / 1e9,
            "mean_launch_us": t * 1e6 / (n_layers * len(groups)),
            "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS}}


def time_step_chained(device, gen, n_layers=4):
    """The headline step with the data dependencies of a decoder layer honoured and the caller's elementwise ops between the
    projections INCLUDED (the reference's layer: integration/BitNet/modeling_bitnet.py:839-860, MLP :240-244): per layer
    {q,k,v}(norm1(x)) -> o_proj(v) + x -> silu(gate(norm2(h))) * up(norm2(h)) -> down_proj(act) + h -> next layer's x (the v
    projection stands in for the attention output: attention and rope are the caller's kernels either way).  `fused`: the
    RMSNorms ride in the activation staging of the group / pair launches, the residual adds in the GEMVs' stores
    (`Matmul.forward_ex`), gate / up / activation are one launch (`matmul_gate_up`) - 4 launches per layer, nothing between
    them; `composed`: the same projections with torch's own kernels around them (`F.rms_norm`, add, silu, mul: 10 launches per
    layer).  One hipGraph replay each, same weights."""
    layers = [[make_linear(N, K, device, gen) for (_, N, K) in LLAMA2_7B_LINEARS] for _ in range(n_layers)]
    for layer in layers:
        for (_, _, sc, _) in layer:
            # random int4 codes have mean -0.5, so every synthetic linear has a common-mode gain of -0.5 * mean(scale) * K: scaled down
            # until that is < 1 and the chained hidden state stays finite over the layers (timing does not depend on the values)
            sc.mul_(0.04)
    x0 = (torch.rand((1, 4096), device=device, generator=gen) - 0.5).to(torch.float16)
    hid = [torch.empty((1, 4096), dtype=torch.float16, device=device) for _ in range(2 * n_layers)]
    act = torch.empty((1, 11008), dtype=torch.float16, device=device)
    norms = [((1.0 + (torch.rand(4096, device=device, generator=gen) - 0.5) * 0.2).to(torch.float16),
              (1.0 + (torch.rand(4096, device=device, generator=gen) - 0.5) * 0.2).to(torch.float16)) for _ in range(n_layers)]
    eps = 1e-5

    def run(fused):
        x = x0
        for li, layer in enumerate(layers):
            h, x_next = hid[2 * li], hid[2 * li + 1]
            q, k, v, o, gate, up, down = layer
            w1, w2 = norms[li]
            if fused:
                bitblas.matmul_group([q[0], k[0], v[0]], x, [(t[1], t[2]) for t in (q, k, v)], outputs=[t[3] for t in (q, k, v)], norm=(w1, eps))
                o[0].forward_ex(v[3], o[1], scale=o[2], residual=x, output=h)
                bitblas.matmul_gate_up(gate[0], up[0], h, (gate[1], gate[2]), (up[1], up[2]), output=act, norm=(w2, eps))
                down[0].forward_ex(act, down[1], scale=down[2], residual=h, output=x_next)
            else:
                xn = torch.nn.functional.rms_norm(x, (4096,), w1, eps)
                bitblas.matmul_group([q[0], k[0], v[0]], xn, [(t[1], t[2]) for t in (q, k, v)], outputs=[t[3] for t in (q, k, v)])
                o[0].forward(v[3], o[1], scale=o[2], output=h)
                h += x
                hn = torch.nn.functional.rms_norm(h, (4096,), w2, eps)
                bitblas.matmul_group([gate[0], up[0]], hn, [(t[1], t[2]) for t in (gate, up)], outputs=[gate[3], up[3]])
                torch.mul(torch.nn.functional.silu(gate[3]), up[3], out=act)
                down[0].forward(act, down[1], scale=down[2], output=x_next)
                x_next += h
            x = x_next
        return x

    # algorithmic bytes of the fused step: the gate / up outputs never reach memory, the activation and two residuals do
    nbytes = n_layers * (sum(algorithmic_bytes(1, N, K) for (_, N, K) in LLAMA2_7B_LINEARS) - 11008 * 2 + 2 * 4096 * 2)
    res = {}
    outs = {}
    for name, fused in (("fused", True), ("composed", False)):
        t = graph_time(device, lambda: run(fused), 1)
        outs[name] = run(fused).float().clone()
        launches = n_layers * (4 if fused else 10)
        res[name] = {"us_per_step": t * 1e6, "launches_per_step": launches, "GBps": nbytes / t / 1e9, "frac": nbytes / t / 1e9 / HBM_PEAK_GBS}
    torch.cuda.synchronize(device)
    ref = outs["composed"]
    err = ((outs["fused"] - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
    from bitblas_amd import gate_up_plan
    return {"workload": f"W_int4 A_fp16 M=1 decode, Llama-2-7B linears, {n_layers} layers CHAINED through their data (o_proj reads v, "
                        "gate/up read o_proj + residual, down_proj reads silu(gate) * up, the next layer reads down_proj + residual), "
                        "the layer's RMSNorms and elementwise ops included; weights as in the headline step",
            "gate_up_launch": (gate_up_plan(layers[0][4][0], 1, norm=True) or {}).get("name"),
            "bytes_per_step": nbytes, **res, "fused_vs_composed_max_rel_err": err,
            "bit_identical": bool(torch.equal(outs["fused"], ref)),
            "roofline": {"bound": "hbm", "achieved": res["fused"]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": res["fused"]["frac"]}}


LLAMA_70B_LINEARS = [(8192, 8192), (28672, 8192), (8192, 28672), (10240, 8192)]   # o, gate / up, down, q+k+v (64 q + 8 + 8 kv heads)


def time_member_group(device, gen, Ns, K, n_sets=None):
    """Projections that share an input as ONE launch (wqaa_matmul_group) at a 70B layer's widths: q/k/v under grouped-query
    attention (8192 + 1024 + 1024 rows) and gate/up (2 x 28672), every member with its own packed tensors and output."""
    ops = [get_op(1, N, K) for N in Ns]
    wbytes = sum(N * K // 2 for N in Ns)
    n_sets = n_sets or max(3, min(32, (640 << 20) // wbytes))
    sets = []
    for _ in range(n_sets):
        sets.append([(torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=device, generator=gen),
                      (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)) for N in Ns])
    outs = [torch.empty((1, N), dtype=torch.float16, device=device) for N in Ns]
    A = (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)

    def launch_all():
        for ws in sets:
            bitblas.matmul_group(ops, A, ws, outputs=outs)

    t = graph_time(device, launch_all, n_sets)
    nbytes = sum(algorithmic_bytes(1, N, K) for N in Ns) - (len(Ns) - 1) * K * 2      # the shared input is read once
    from bitblas_amd import group_plan
    gp = group_plan(ops, 1)
    return {"workload": f"W_int4 A_fp16 GEMV M=1, {len(Ns)} projections N={list(Ns)} K={K} g=128 sharing one input",
            "kernel": (gp["plan"] or {}).get("name"), "launches": gp["launches"],
            "us_per_launch": t * 1e6, "bytes_per_launch": nbytes, "GBps": nbytes / t / 1e9,
            "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS}}


def time_member_f16_gemv(device, gen, N, K, int4_us=None):
    """The reference's published figure for this path is a SPEED-UP over the vendor library's float16 GEMV (README.md:43-48,
    images/figures/op_benchmark_a100_wq_gemv_e7.png: W_INT4 A_FP16 M = 1 about 3.9-4.3x cuBLAS on A100).  Same
    yardstick here: the float16 weight of the same shape through torch.matmul (the vendor library on this box) and through this
    library's own dense float16 GEMV, hipGraph replays over rotating weights like the quantised members."""
    try:
        n_buf = max(4, min(32, (640 << 20) // (N * K * 2)))
        Ws = [(torch.rand((N, K), device=device, generator=gen) - 0.5).to(torch.float16) for _ in range(n_buf)]
        A = (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)
        out = torch.empty((1, N), dtype=torch.float16, device=device)
        op = bitblas.Matmul(bitblas.MatmulConfig(M=1, N=N, K=K, A_dtype="float16", W_dtype="float16", accum_dtype="float16",
                                                 out_dtype="float16"), enable_tuning=False)

        def launch_own():
            stream = torch.cuda.current_stream(device).cuda_stream
            for W in Ws:
                op.lib.run(A.data_ptr(), W.data_ptr(), None, None, None, None, out.data_ptr(), 1, stream)

        def launch_vendor():
            for W in Ws:
                torch.matmul(A, W.t(), out=out)

        t_own = graph_time(device, launch_own, n_buf)
        t_vendor = graph_time(device, launch_vendor, n_buf)
        nbytes = N * K * 2 + K * 2 + N * 2
        res = {"workload": f"float16 x float16 GEMV M=1 N={N} K={K} ({nbytes >> 20} MiB per launch)",
               "own_us_per_launch": t_own * 1e6, "own_GBps": nbytes / t_own / 1e9, "own_kernel": op.plans[1]["name"],
               "vendor_us_per_launch": t_vendor * 1e6, "vendor_GBps": nbytes / t_vendor / 1e9, "vendor": "torch.matmul (rocBLAS / hipBLASLt)"}
        if int4_us:
            res["int4_speedup_vs_vendor_f16"] = t_vendor * 1e6 / int4_us
            res["int4_speedup_vs_own_f16"] = t_own * 1e6 / int4_us
            res["reference_published"] = "W_INT4 A_FP16 GEMV about 3.9-4.3x cuBLAS float16 on A100 (chart, other hardware)"
        return res
    except Exception as exc:  # yardstick not available: report, never fake
        return {"error": f"{type(exc).__name__}: {exc}"}


def time_member_resident_decode(device, gen, M=4096, N=4096, K=4096, n_buf=4):
    """`Linear.enable_decoded_weight_cache` (bitblas_amd/module.py): the TE graph's B_decode kept resident in HBM (N*K*2 bytes
    per layer) and the plain dense GEMM run against it - the two-pass member with its first pass hoisted out of the call.
    Same operands as `gemm_uint4_m4096`; the decode happens once, in the untimed first call."""
    try:
        lins = []
        for _ in range(n_buf):
            lin = bitblas.Linear(K, N, A_dtype="float16", W_dtype="uint4", group_size=GROUP, with_scaling=True, with_zeros=True,
                                 zeros_mode="original", opt_M=[16, M], enable_tuning=False).to(device)
            lin.qweight = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=device, generator=gen)
            lin.scales = (torch.rand((N, K // GROUP), device=device, generator=gen) * 0.02).to(torch.float16)
            lin.zeros = torch.full((N, K // GROUP), 8.0, dtype=torch.float16, device=device)
            lin.enable_decoded_weight_cache(min_m=256)
            lins.append(lin)
        lins[0]._dense_op.hardware_aware_finetune()        # the vendor library's candidates for the dense pair, timed (shared operator)
        A = (torch.rand((M, K), device=device, generator=gen) - 0.5).to(torch.float16)
        out = torch.empty((M, N), dtype=torch.float16, device=device)

        def launch_all():
            for lin in lins:
                lin(A, output=out)

        t = graph_time(device, launch_all, n_buf)
        tf = 2.0 * M * N * K / t / 1e12
        return {"workload": f"W_uint4 A_float16 GEMM M={M} N={N} K={K} g=128 zeros=original, B_decode resident ({N * K * 2 >> 20} MiB per layer)",
                "kernel": lins[0]._dense_op.plans[M]["name"] if M in lins[0]._dense_op.plans else lins[0]._dense_op.lib.plan(M)["name"],
                "us_per_launch": t * 1e6, "TFLOPs": tf, "frac_of_mfma_peak": tf / MFMA_F16_PEAK_TF, "mfma_peak": MFMA_F16_PEAK_TF,
                "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TF,
                             "flops_per_launch": 2.0 * M * N * K}}
    except Exception as exc:  # member not available: report, never fake
        return {"error": f"{type(exc).__name__}: {exc}"}


def time_member_dense(device, gen, M, N, K, kind="fp8", n_buf=4, vendor=False, tuned=False):
    """Dense members: e4m3 x e4m3 MFMA GEMM on Llama-3-70B shapes (BASELINE config c5, one GPU's unsharded
    matrix) and the M = 1 W_int2 A_int8 GEMV (c4)."""
    import bitblas_amd as bitblas
    # vendor=True: the YARDSTICK - hipBLASLt's GEMM on the same operands (csrc/wqaa_dense_lib.hip).  The product runs this
    # library's own kernels; WQAA_DENSE_LIB=1 is the opt-in, a plan-time switch: set while the operator is planned AND timed
    if vendor:
        os.environ["WQAA_DENSE_LIB"] = "1"
    op = None
    try:
        if kind == "fp8":
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32",
                                       out_dtype="float16")
        elif kind == "f16":
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="float16", W_dtype="float16", accum_dtype="float32", out_dtype="float16")
        elif kind == "int8":
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int8", W_dtype="int8", accum_dtype="int32", out_dtype="int32")
        else:
            cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="int8", W_dtype="int2", accum_dtype="int32", out_dtype="int32")
        op = bitblas.Matmul(cfg, enable_tuning=False)
        if tuned:
            op.hardware_aware_finetune()       # the vendor library's candidate algorithms timed on the device (wqaa_tune)
        return _time_member_dense(device, gen, op, M, N, K, kind, n_buf)
    except Exception as exc:  # member not built: report, never fake
        return {"error": str(exc)}
    finally:
        if vendor:
            del os.environ["WQAA_DENSE_LIB"]
            if op is not None:
                op.lib.plan(M)                 # planning re-reads the switch for whatever runs next


def _time_member_dense(device, gen, op, M, N, K, kind, n_buf):
    if kind == "fp8":
        A = (torch.rand((M, K), device=device, generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
        Ws = [(torch.rand((N, K), device=device, generator=gen) * 2 - 1).to(torch.float8_e4m3fn) for _ in range(n_buf)]
        out = torch.empty((M, N), dtype=torch.float16, device=device)
        wbytes = N * K
    elif kind == "f16":
        A = (torch.rand((M, K), device=device, generator=gen) - 0.5).to(torch.float16)
        Ws = [(torch.rand((N, K), device=device, generator=gen) - 0.5).to(torch.float16) for _ in range(n_buf)]
        out = torch.empty((M, N), dtype=torch.float16, device=device)
        wbytes = N * K * 2
    elif kind == "int8":
        A = torch.randint(-128, 128, (M, K), device=device, dtype=torch.int8, generator=gen)
        Ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=device, generator=gen) for _ in range(n_buf)]
        out = torch.empty((M, N), dtype=torch.int32, device=device)
        wbytes = N * K
    else:
        A = torch.randint(-128, 128, (M, K), device=device, dtype=torch.int8, generator=gen)
        Ws = [torch.randint(-128, 128, (N, K // 4), dtype=torch.int8, device=device, generator=gen) for _ in range(n_buf)]
        out = torch.empty((M, N), dtype=torch.int32, device=device)
        wbytes = N * K // 4

    def launch_all():
        stream = torch.cuda.current_stream(device).cuda_stream
        for W in Ws:
            op.lib.run(A.data_ptr(), W.data_ptr(), None, None, None, None, out.data_ptr(), M, stream)

    t = graph_time(device, launch_all, n_buf)
    peak = MFMA_F16_PEAK_TF if kind == "f16" else MFMA_I8_PEAK_TOPS
    res = {"workload": f"{'e4m3 x e4m3' if kind == 'fp8' else 'float16 x float16' if kind == 'f16' else 'int8 x int8' if kind == 'int8' else 'W_int2 A_int8'} M={M} N={N} K={K}",
           "kernel": op.plans[M]["name"], "us_per_launch": t * 1e6}
    if M == 1:
        nbytes = M * K + wbytes + M * N * out.element_size()
        res.update(bytes_per_launch=nbytes, GBps=nbytes / t / 1e9, frac_of_hbm_peak=nbytes / t / 1e9 / HBM_PEAK_GBS,
                   roofline={"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": nbytes / t / 1e9 / HBM_PEAK_GBS})
    else:
        tf = 2.0 * M * N * K / t / 1e12
        res.update(TFLOPs=tf, frac_of_mfma_peak=tf / peak, mfma_peak=peak,
                   roofline={"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                             "flops_per_launch": 2.0 * M * N * K})
    return res


def time_c5_sharded(device, gen, world, rank, steps=5, warmup=2):
    """BASELINE config c5 as the multi-GPU workload: dense e4m3 x e4m3, M = 4096, the four Llama-3-70B linears, N column-
    sharded over the ranks (rank p owns rows [p N/P, (p+1) N/P) of W), output slices all-gathered over RCCL in row blocks
    (ColumnParallelMatmul.auto_row_block) under the next block's GEMM (bitblas_amd/parallel.py).  Strong scaling: the total work is fixed."""
    import torch.distributed as dist
    from bitblas_amd.parallel import ColumnParallelMatmul
    M = 4096
    shapes = [("o_proj", 8192, 8192), ("down_proj", 8192, 28672), ("qkv_proj", 10240, 8192), ("gate_proj", 28672, 8192)]
    ops = []
    for (_, N, K) in shapes:
        cfg = bitblas.MatmulConfig(M=M, N=N, K=K, A_dtype="e4m3_float8", W_dtype="e4m3_float8", accum_dtype="float32", out_dtype="float16")
        op = ColumnParallelMatmul(cfg)
        A = (torch.rand((M, K), device=device, generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
        W = (torch.rand((N // world, K), device=device, generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
        out = torch.empty((M, N), dtype=torch.float16, device=device)
        ops.append((op, A, W, out))

    def step():
        for (op, A, W, out) in ops:
            op(A, W, out=out)

    for _ in range(warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    flops = sum(2.0 * M * N * K for (_, N, K) in shapes)
    return {"workload": "e4m3 x e4m3 GEMM M=4096, Llama-3-70B o/down/qkv/gate, N column-sharded over the ranks, all-gather of the "
                        "[4096, N/P] float16 slices in row blocks under the next block's GEMM",
            "row_blocks": [op.row_block for (op, _, _, _) in ops],
            "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "TFLOPs_whole_job": flops * steps / elapsed / 1e12,
            "gathered_bytes_per_step_per_rank": sum(M * N * 2 for (_, N, _) in shapes) * (world - 1) // max(world, 1),
            "kernels": [op.op.plans[M]["name"] if M in op.op.plans else None for (op, _, _, _) in ops]}


def usable_cores():
    """host cores this process may actually run on: the affinity mask and the cgroup CPU quota, not the machine's core
    count (256 OpenMP threads on a 16-CPU quota measure the scheduler: 544 ms for a 16 M element dequantise)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().strip())))
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(max_seconds=20.0):
    """The reference's CPU path restated: dequantise the int4 weights to float16 values and take
    the fp32 matmul, timed on ALL host cores - both stages in torch (threaded); the first pass is checked against the
    numpy oracle (oracle/wqaa_oracle.py), which is single-threaded and would measure numpy, not the host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wqaa_oracle as oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    N = K = 4096
    A = (rng.random((1, K), dtype=np.float32) - 0.5).astype(np.float16)
    codes = rng.integers(0, 16, size=(N, K)).astype(np.int8)
    scale = (rng.random((N, K // GROUP), dtype=np.float32) * 0.02).astype(np.float16)
    At, Ct, St = torch.from_numpy(A), torch.from_numpy(codes), torch.from_numpy(scale)

    S32 = St.float().repeat_interleave(GROUP, dim=1)

    def dequant():
        # (w - 8) * s rounded to float16 per element, as the TE definition does: w - 8 and the product of a 4-bit by an
        # 11-bit significand are exact in fp32, so one cast to float16 is the definition's single rounding
        return ((Ct.float() - 8.0) * S32).half()

    Wd = dequant()
    want = oracle.dequantize_weight(codes, "int", 4, K=K, scale=scale, group_size=GROUP)
    assert np.array_equal(Wd.numpy().view(np.uint16), np.asarray(want, dtype=np.float16).view(np.uint16)), "torch dequant != oracle"
    t0 = time.perf_counter()
    n = 0
    deq_t = mm_t = 0.0
    while True:
        t1 = time.perf_counter()
        Wd = dequant()
        t2 = time.perf_counter()
        torch.matmul(At.float(), Wd.float().T).half()
        t3 = time.perf_counter()
        deq_t += t2 - t1
        mm_t += t3 - t2
        n += 1
        if time.perf_counter() - t0 > max_seconds or n >= 200:
            break
    per = (deq_t + mm_t) / n
    nbytes = algorithmic_bytes(1, N, K)
    return {"value": nbytes / per / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{n} x (torch-threaded dequantise + fp32 matmul) of W_int4 A_fp16 M=1 N=K=4096 g=128, first pass checked "
                      f"bit for bit against the numpy oracle (dequant {deq_t / n * 1e3:.1f} ms + matmul {mm_t / n * 1e3:.1f} ms per pass)"}


def pmc_traffic(launches_per_step):
    """HBM bytes per launch of the step from the newest committed rocprofv3 --pmc run (profiles/*pmc*.json), or None.
    The counters are collected per step (tools/profile_round.sh) and divided by this run's launches per step."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("gemv_hbm_bytes_per_step") is not None:
                return d["gemv_hbm_bytes_per_step"] / launches_per_step
            if d.get("gemv_hbm_bytes_per_launch") is not None:      # files from before the group launches: 28 launches per step
                return d["gemv_hbm_bytes_per_launch"] * 28 / launches_per_step
        except Exception:
            continue
    return None


def pmc_traffic_live(launches_per_step, layers, timeout_s=240):
    """HBM bytes per launch of the step, measured NOW: two child runs of this file's step (eager launches, no members) under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` - separate passes, counters only, as MI355X_MICROARCH.md's HBM section
    prescribes - reduced with its gfx950 corrections (both in KiB; a wide streaming read is counted at half its bytes).
    Returns (bytes per launch, detail) or (None, reason): never raises, never runs nested under a profiler."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROFILER_", "ROCP_TOOL", "ROCPROF_")) for k in os.environ):
        return None, "already under rocprofv3"
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None, "rocprofv3 not found"
    kib = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [tool, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--layers", str(layers), "--no-cpu-baseline", "--no-members", "--eager", "--no-live-pmc"]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            except Exception as exc:  # noqa: BLE001
                return None, f"{ctr} pass: {type(exc).__name__}"
            vals = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        name = row.get("Kernel_Name", "")
                        if row.get("Counter_Name") == ctr and ("wq_gemv_kernel" in name or "wq_gemvx_kernel" in name):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"{ctr} pass: no GEMV launches in the counter file"
            kib[ctr] = (sum(vals) / len(vals), len(vals))
    read = kib["FETCH_SIZE"][0] * 1024 * 2
    write = kib["WRITE_SIZE"][0] * 1024
    return read + write, {"source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this step (eager launches), "
                                    "gfx950 corrections of MI355X_MICROARCH.md", "read_bytes_per_launch": read,
                          "write_bytes_per_launch": write, "launches_counted": kib["FETCH_SIZE"][1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-members", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profile instead of a live counter pass")
    ap.add_argument("--eager", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-groups", action="store_true", help="7 launches per layer instead of {q,k,v}, o, {gate,up}, down")
    ap.add_argument("--members-out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "bench_members.json"),
                    help="side file for the per-member records (the last stdout line carries the contract fields only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # WQAA_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL init, per-step all-gather) with one rank - the only way to
    # exercise it on a 1-GPU box
    dist_on = world > 1 or os.environ.get("WQAA_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)

    # ---- resident synthetic model shard: every rank holds LAYERS x 7 linears (its N-slice) ----
    layers = []
    for _ in range(args.layers):
        layers.append([make_linear(N, K, device, gen) for (_, N, K) in LLAMA2_7B_LINEARS])
    acts = {K: (torch.rand((1, K), device=device, generator=gen) - 0.5).to(torch.float16)
            for K in (4096, 11008)}
    step_bytes = args.layers * sum(algorithmic_bytes(1, N, K) for (_, N, K) in LLAMA2_7B_LINEARS)
    # launch structure of a layer: the projections that share an input form one group (one launch)
    LAYER_GROUPS = [[0], [1], [2], [3], [4], [5], [6]] if args.no_groups else [[0, 1, 2], [3], [4, 5], [6]]
    launches_per_step = args.layers * len(LAYER_GROUPS)

    # N > 1: every rank writes its column slice of a step into one of TWO staging buffers and the RCCL all-gather
    # of step i runs on RCCL's stream while step i+1 computes into the other buffer (a buffer is reused only after
    # its gather has been waited for) - throughput is max(compute, gather) per step, not the sum
    gathered = local_out = None
    n_slots = 2 if dist_on else 1
    if dist_on:
        flat_n = sum(N for (_, N, _) in LLAMA2_7B_LINEARS)
        local_out = [torch.empty((args.layers, flat_n), dtype=torch.float16, device=device) for _ in range(n_slots)]
        gathered = [torch.empty((world * args.layers, flat_n), dtype=torch.float16, device=device) for _ in range(n_slots)]

    def launch_layers(slot, groups=None):
        groups = LAYER_GROUPS if groups is None else groups
        stream = torch.cuda.current_stream(device).cuda_stream
        for li, layer in enumerate(layers):
            offs = np.cumsum([0] + [op.N for (op, _, _, _) in layer])
            dsts = [out if not dist_on else local_out[slot][li:li + 1, offs[i]:offs[i] + op.N]
                    for i, (op, _, _, out) in enumerate(layer)]
            for grp in groups:
                if len(grp) == 1:
                    op, qw, sc, _ = layer[grp[0]]
                    op.lib.run(acts[op.K].data_ptr(), qw.data_ptr(), None, sc.data_ptr(), None, None,
                               dsts[grp[0]].data_ptr(), 1, stream)
                else:
                    ops = [layer[i][0] for i in grp]
                    bitblas.matmul_group(ops, acts[ops[0].K], [(layer[i][1], layer[i][2]) for i in grp],
                                         outputs=[dsts[i] for i in grp])

    # The collective rides in the hipGraph: the graph of a step forks a side stream that all-gathers the PREVIOUS
    # step's staging buffer while the main branch runs this step's GEMVs into the other one, then joins.  One graph
    # replay per step, no per-step host call into RCCL (an eager all_gather_into_tensor costs ~50 us of host time
    # against a 160 us step).  If RCCL refuses capture the eager double-buffered scheme below takes over.
    graphs = None
    gather_in_graph = False
    pending = [None] * n_slots
    step_no = [0]

    # WQAA_BENCH_GATHER: "serial" (default) = the step's graph ends with the all-gather of its own staging buffer, one
    # linear graph; "overlap" = the graph forks a side stream that gathers the PREVIOUS step's buffer while this
    # step's GEMVs run (measured slower at one rank: a forked hipGraph loses 70 us per replay); "eager" = no capture
    gather_mode = os.environ.get("WQAA_BENCH_GATHER", "serial") if dist_on else "none"
    if dist_on and args.eager:
        gather_mode = "eager"

    def capture(mode):
        out = []
        if mode in ("serial", "overlap"):
            import torch.distributed as dist
            for slot in range(n_slots):   # communicator set-up and buffer registration happen outside capture
                dist.all_gather_into_tensor(gathered[slot], local_out[slot])
            torch.cuda.synchronize(device)
        side = torch.cuda.Stream(device=device) if mode == "overlap" else None
        for slot in range(n_slots):
            launch_layers(slot)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                if mode == "overlap":
                    main = torch.cuda.current_stream(device)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        dist.all_gather_into_tensor(gathered[1 - slot], local_out[1 - slot])
                launch_layers(slot)
                if mode == "overlap":
                    main.wait_stream(side)
                if mode == "serial":
                    dist.all_gather_into_tensor(gathered[slot], local_out[slot])
            out.append(g)
        return out

    if not args.eager:
        if gather_mode in ("serial", "overlap"):
            try:
                graphs = capture(gather_mode)
                gather_in_graph = True
            except Exception as e:  # noqa: BLE001 - any capture failure: fall back, say so
                print(f"[bench] rank {rank}: RCCL all-gather not capturable ({type(e).__name__}: {e}); eager gather",
                      file=sys.stderr)
                torch.cuda.synchronize(device)
                graphs = None
                gather_mode = "eager"
        if graphs is None:
            graphs = capture("none")
    graph = graphs[0] if graphs else None

    def one_step():
        slot = step_no[0] % n_slots
        step_no[0] += 1
        if pending[slot] is not None:
            pending[slot].wait()          # stream-side wait: the staging buffer is free again
            pending[slot] = None
        if graphs is not None:
            graphs[slot].replay()
        else:
            launch_layers(slot)
        if dist_on and not gather_in_graph:
            import torch.distributed as dist
            pending[slot] = dist.all_gather_into_tensor(gathered[slot], local_out[slot], async_op=True)

    def drain():
        if gather_in_graph and gather_mode == "serial":
            return
        if gather_in_graph:
            # the last step's slice has not been gathered by a following replay yet
            import torch.distributed as dist
            last = (step_no[0] - 1) % n_slots
            dist.all_gather_into_tensor(gathered[last], local_out[last])
            return
        for slot in range(n_slots):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    drain()
    barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step()
    drain()                               # every step's gather is inside the timed region
    ev1.record()
    torch.cuda.synchronize(device)
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_elapsed = ev0.elapsed_time(ev1) * 1e-3
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        value = step_bytes * args.steps * world / elapsed / 1e9
        # dominant kernel: all launches of a step are one kernel function; average duration from the
        # events bracketing the timed replays on the launch stream (all-gather time excluded at N=1)
        avg_launch_s = gpu_elapsed / (args.steps * launches_per_step)
        avg_bytes = step_bytes / launches_per_step
        achieved = avg_bytes / avg_launch_s / 1e9
        kernel_name = layers[0][0][0].plans[1]["name"].replace("m1n4096k4096", "m1")
        from bitblas_amd import group_plan
        group_names = [("+".join(LLAMA2_7B_LINEARS[i][0] for i in grp),
                        (group_plan([layers[0][i][0] for i in grp], 1)["plan"] or layers[0][grp[0]][0].plans[1])["name"])
                       for grp in LAYER_GROUPS]
        result = {
            "metric": "achieved HBM GB/s of the W_int4 A_fp16 GEMV at M=1, Llama-2-7B linear shapes, g=128 "
                      "(+ TFLOP/s of the M=4096 MFMA GEMM under `members`)",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"W_int4 A_fp16 GEMV M=1, Llama-2-7B linears (N,K in {{4096,11008}}), g=128: "
                                   f"{args.layers} layers x 7 GEMV per step per GPU"
                                   + (", 7 launches per layer, " if args.no_groups else
                                      " in 4 launches per layer ({q,k,v}, o, {gate,up}, down: wqaa_matmul_group runs the projections "
                                      "that share an input as one launch, each with its own packed tensors and output), ")
                                   + f"{'one hipGraph replay per step' if graph is not None else 'eager launches'}",
                       "numerics": NUMERICS,
                       "launches": {k: v for k, v in group_names},
                       "launches_per_step": launches_per_step, "bytes_per_step_per_gpu": step_bytes,
                       "sharding": (f"column (N) shard per rank + 1 RCCL all-gather per step ({gather_mode}: " +
                                    {"serial": "captured at the end of the step's hipGraph",
                                     "overlap": "captured on a forked branch of the next step's hipGraph",
                                     "eager": "eager async call, overlapped with the next step, two staging buffers"
                                     }.get(gather_mode, "") + ")")
                       if dist_on else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(launches_per_step),
                         "kernel": "wq_gemvx_kernel<int4, lop3, scale, mb1> (every launch of the step: " +
                                   ", ".join(sorted({v for _, v in group_names})) + ")",
                         "numerics": "strict_reference=False: exact products, group scale on fp32 partial sums (1e-3 contract vs the "
                                     "reference definition: tests/test_gemvx_gpu.py); the per-element-rounding members are timed "
                                     "under members[*_strict]",
                         "bytes_per_launch": avg_bytes, "mean_launch_us": avg_launch_s * 1e6,
                         "timing": "torch.cuda.Event pair on the launch stream around the timed graph replays / "
                                   "(steps x launches per step); weights rotate over 420 MB per step"},
        }
        if not args.no_members and world == 1:
            members = {}

            def member(name, fn, *a, **k):
                """one member = one try: a member that cannot run (or a capture that fails) is reported under its name and
                the JSON line still comes out with everything else"""
                try:
                    members[name] = fn(*a, **k)
                except Exception as exc:  # noqa: BLE001
                    members[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    try:
                        torch.cuda.synchronize(device)
                    except Exception:  # noqa: BLE001
                        pass

            def step_ungrouped():
                # the same step with every GEMV as its own launch (7 per layer): what the grouping buys
                flat = [[i] for i in range(len(LLAMA2_7B_LINEARS))]
                t_step = graph_time(device, lambda: launch_layers(0, flat), 1)
                return {"workload": "the headline step, 7 launches per layer", "us_per_step": t_step * 1e6,
                        "GBps": step_bytes / t_step / 1e9, "frac_of_hbm_peak": step_bytes / t_step / 1e9 / HBM_PEAK_GBS,
                        "roofline": {"bound": "hbm", "achieved": step_bytes / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": step_bytes / t_step / 1e9 / HBM_PEAK_GBS}}

            if not args.no_groups:
                member("step_ungrouped", step_ungrouped)
            for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096)):     # c2 shapes
                member(f"gemv_int4_n{N}k{K}", time_member_gemv, device, gen, N, K)
                member(f"gemv_int4_n{N}k{K}_strict", time_member_gemv, device, gen, N, K, strict=True)
            # what BASELINE.json's `metric` is quoted on - W_int4 A_fp16 at M = 1 and M = 4096 on the Llama-70B linears (the
            # reference's benchmark table, benchmark/README.md:60-62 V10-V12 and :73-75 M10-M12) - plus the q/k/v width of
            # grouped-query attention and the group launches of a 70B layer
            for (N, K) in LLAMA_70B_LINEARS:
                member(f"gemv_int4_n{N}k{K}", time_member_gemv, device, gen, N, K)
            for (N, K) in LLAMA_70B_LINEARS[1:3]:
                member(f"gemv_int4_n{N}k{K}_strict", time_member_gemv, device, gen, N, K, strict=True)
            member("group_int4_70b_qkv", time_member_group, device, gen, (8192, 1024, 1024), 8192)
            member("group_int4_70b_gate_up", time_member_group, device, gen, (28672, 28672), 8192)
            for (N, K) in LLAMA_70B_LINEARS[:3]:
                member(f"gemm_uint4_m4096_n{N}k{K}", time_member_gemm, device, gen, 4096, N, K, n_buf=2)
            for (N, K) in ((4096, 4096), (11008, 4096)):        # the reference's own yardstick: speed-up over the float16 GEMV
                member(f"gemv_f16_yardstick_n{N}k{K}", time_member_f16_gemv, device, gen, N, K,
                       int4_us=(members.get(f"gemv_int4_n{N}k{K}") or {}).get("us_per_launch"))
            member("gemm_uint4_m4096", time_member_gemm, device, gen, 4096)
            member("gemm_uint4_m4096_two_pass_vendor", time_member_gemm, device, gen, 4096, tuned=True)
            member("gemm_uint4_m4096_fractional_zeros", time_member_gemm, device, gen, 4096, frac_zeros=True)
            member("gemm_uint4_m128", time_member_gemm, device, gen, 128)
            member("gemm_uint4_m16", time_member_gemm, device, gen, 16)
            # the rest of the reference's default opt_M steps below the ping-pong tiles ("M = 32 / 64 / 256 reported"), and a
            # shape where the mid-M member (round 5, `...xmk`) is the selector's choice below 65 rows
            for m_ in (32, 64, 256):
                member(f"gemm_uint4_m{m_}", time_member_gemm, device, gen, m_)
            member("gemm_uint4_m64_n4096k8192", time_member_gemm, device, gen, 64, 4096, 8192)
            # decode batches on wide outputs (round 4): the persistent form of the one-launch decode member (a 7B model's gate / up and
            # q/k/v widths, `...xdlp`) and its whole-tile form on long K (a 70B model's hidden size, `...xdlt`)
            member("gemm_uint4_m8_n11008k4096", time_member_gemm, device, gen, 8, 11008, 4096)
            member("gemm_uint4_m8_n22016k4096", time_member_gemm, device, gen, 8, 22016, 4096)
            member("gemm_uint4_m8_n8192k8192", time_member_gemm, device, gen, 8, 8192, 8192)
            # long K (round 5): a 7B model's down projection (one-launch form) and a 70B model's, where the K-sliced form of
            # the decode member (`...xdlk`) is the selector's choice
            member("gemm_uint4_m8_n4096k11008", time_member_gemm, device, gen, 8, 4096, 11008)
            member("gemm_uint4_m8_n8192k28672", time_member_gemm, device, gen, 8, 8192, 28672)
            member("gemm_uint4_m16_n8192k28672", time_member_gemm, device, gen, 16, 8192, 28672)
            # a vocabulary projection at a decode batch: wider than the persistent form reaches - a wave per fragment (`...xdlw`, round 5)
            member("gemm_uint4_m8_n32000k4096", time_member_gemm, device, gen, 8, 32000, 4096)
            member("gemm_int2_int8_m4096", time_member_gemm, device, gen, 4096, W_dtype="int2", A_dtype="int8")
            member("gemm_int2_int8_m4096_bitnet", time_member_gemm, device, gen, 4096, W_dtype="int2", A_dtype="int8", bitnet=True)
            # the reference's plain matmul (float16 x float16, README.md support matrix): this library's dense member on the ping-pong
            # skeleton (round 4) and, as a yardstick, the vendor library on the same operands
            member("gemm_f16_dense_m4096", time_member_dense, device, gen, 4096, 4096, 4096, kind="f16", n_buf=4)
            member("gemm_f16_dense_m4096_vendor", time_member_dense, device, gen, 4096, 4096, 4096, kind="f16", n_buf=4, vendor=True, tuned=True)
            member("gemm_int8_dense_m4096", time_member_dense, device, gen, 4096, 4096, 4096, kind="int8", n_buf=4)
            member("gemv_int2_int8_m1", time_member_dense, device, gen, 1, 4096, 4096, kind="int2", n_buf=64)
            member("step_chained", time_step_chained, device, gen)
            member("step_int2_int8", time_step_int2_int8, device, gen)
            member("step_int2_int8_ungrouped", time_step_int2_int8, device, gen, grouped=False)
            # c5: dense e4m3 x e4m3 on every Llama-3-70B linear of one (unsharded) GPU, M = 4096 and M = 1
            # (this library's own ping-pong MFMA member; `_vendor` = the hipBLASLt yardstick on the same operands, tuned)
            for (name, N, K, nb) in (("o", 8192, 8192, 4), ("down", 8192, 28672, 2), ("qkv", 10240, 8192, 4), ("gate", 28672, 8192, 2)):
                member(f"gemm_fp8_m4096_{name}_n{N}_k{K}", time_member_dense, device, gen, 4096, N, K, n_buf=nb)
                member(f"gemm_fp8_m4096_{name}_n{N}_k{K}_vendor", time_member_dense, device, gen, 4096, N, K, n_buf=nb, vendor=True, tuned=True)
            for (name, N, K) in (("o", 8192, 8192), ("down", 8192, 28672)):
                member(f"gemv_fp8_m1_{name}_n{N}_k{K}", time_member_dense, device, gen, 1, N, K, n_buf=max(3, (640 << 20) // (N * K)))
            result["members"] = members
        result["roofline"]["traffic_source"] = "committed rocprofv3 --pmc run under profiles/ (tools/profile_round.sh)"
        if not args.no_live_pmc and not args.no_members and world == 1:
            live, detail = pmc_traffic_live(launches_per_step, args.layers)
            if live is not None:
                result["roofline"]["traffic"] = live
                result["roofline"]["traffic_source"] = detail
            else:
                result["roofline"]["traffic_source"] += f"; live pass skipped: {detail}"
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # noqa: BLE001 - the line must come out
                result["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    if dist_on and not args.no_members:
        c5 = time_c5_sharded(device, gen, world, rank)          # every rank takes part; rank 0 reports
        if rank == 0:
            result["multi_gpu_c5"] = c5
            result["config"]["workload"] += "; + BASELINE c5 (e4m3 x e4m3 M=4096, N column-sharded, RCCL all-gather) under `multi_gpu_c5`"
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL prints its banner through C stdio, which (redirected) would otherwise be
        # flushed at exit, after Python's own buffer
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        # the driver keeps 8 KB of stdout and could not parse round 4's 24 KB line.  The per-member records go to a
        # side file and to an EARLIER stdout line; the last line of stdout is the contract line alone, < 1.8 KB (final_line,
        # tests/test_bench_accounting.py)
        members = result.pop("members", None)
        if members is not None:
            emit_members(members, args.members_out)
        print(json.dumps(final_line(result, members)), flush=True)


MEMBER_KEYS = ("gemm_uint4_m4096", "gemm_uint4_m128", "gemm_uint4_m16", "gemv_int4_n4096k4096", "gemm_int2_int8_m4096",
               "gemv_int2_int8_m1",
               # the metric's own shapes (Llama-70B linears, reference benchmark/README.md:60-62, 73-75)
               "gemv_int4_n8192k28672", "gemv_int4_n28672k8192", "gemm_uint4_m4096_n28672k8192")


def brief_member(v):
    """time and roofline fraction of one member record"""
    if not isinstance(v, dict) or "error" in v:
        return v
    t = v.get("us_per_launch", v.get("us_per_step"))
    if t is None and isinstance(v.get("fused"), dict):
        return {"fused_us": round(v["fused"]["us_per_step"], 1), "composed_us": round(v["composed"]["us_per_step"], 1),
                "frac": round(v["fused"]["frac"], 3)}
    if t is None and "own_us_per_launch" in v:
        return {"own_f16_us": round(v["own_us_per_launch"], 2), "vendor_f16_us": round(v["vendor_us_per_launch"], 2),
                "int4_speedup_vs_vendor_f16": round(v.get("int4_speedup_vs_vendor_f16") or 0.0, 2)}
    frac = (v.get("roofline") or {}).get("frac")
    return {"us": None if t is None else round(t, 2), "frac": None if frac is None else round(frac, 3)}


def emit_members(members, path):
    """the per-member records: one stdout line of their own (NOT the last one) and a side file the round's evidence is copied from"""
    record = {"members": members, "members_summary": {k: brief_member(v) for k, v in members.items()}}
    print("[bench-members] " + json.dumps(record["members_summary"]), flush=True)
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(record, f, indent=1)
        except OSError as exc:
            print(f"[bench] members file not written: {exc}", file=sys.stderr)


def _clip(text, n):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def final_line(result, members=None):
    """The ONE line the driver parses: the contract fields only, every string bounded, < 1.8 KB whatever ran.  `members`: the
    BASELINE-named configurations' time / roofline fraction ride along as {us, frac} pairs (MEMBER_KEYS), nothing else of them."""
    cfg = result.get("config", {})
    roof = result.get("roofline", {})
    cpu = result.get("cpu_baseline")
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                         "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = _clip(line["metric"], 90)
    for k in ("value", "ms_per_step"):
        if isinstance(line.get(k), float):
            line[k] = round(line[k], 4)
    line["config"] = {"workload": _clip(cfg.get("workload", ""), 120), "numerics": _clip(cfg.get("numerics", NUMERICS), 160),
                      "launches_per_step": cfg.get("launches_per_step"),
                      "bytes_per_step_per_gpu": cfg.get("bytes_per_step_per_gpu"), "sharding": _clip(cfg.get("sharding", "none"), 40)}
    line["roofline"] = {"bound": roof.get("bound"), "achieved": None if roof.get("achieved") is None else round(roof["achieved"], 2),
                        "peak": roof.get("peak"), "unit": roof.get("unit"),
                        "frac": None if roof.get("frac") is None else round(roof["frac"], 4), "traffic": roof.get("traffic"),
                        "kernel": _clip(roof.get("kernel", ""), 70), "bytes_per_launch": roof.get("bytes_per_launch"),
                        "mean_launch_us": None if roof.get("mean_launch_us") is None else round(roof["mean_launch_us"], 3)}
    if isinstance(cpu, dict):
        line["cpu_baseline"] = ({"error": _clip(cpu["error"], 120)} if "error" in cpu else
                                {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                 "sample": _clip(cpu.get("sample", ""), 100)})
    if members:
        line["members"] = {k: brief_member(members[k]) for k in MEMBER_KEYS
                           if k in members and not (isinstance(members[k], dict) and "error" in members[k])}
        line["members_file"] = "gpurun_out/bench_members.json"
    if isinstance(result.get("multi_gpu_c5"), dict):
        c5 = result["multi_gpu_c5"]
        line["multi_gpu_c5"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c5.items()
                                if isinstance(v, (int, float)) or (isinstance(v, str) and len(v) <= 40)}
        if len(json.dumps(line["multi_gpu_c5"])) > 400:
            line["multi_gpu_c5"] = {"see": "stderr"}
    return line


if __name__ == "__main__":
    main()
