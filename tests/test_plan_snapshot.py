"""Selector snapshot: the tile configuration `wqaa_select` answers for every BASELINE configuration and bench.py member is
pinned (tests/golden/plan_snapshot.json).  The kernels' speed IS their selection - grid, workgroup width, K split, member
family were each chosen by a same-box A/B - so an edit to the selector that moves one of them has
to show up in a diff, not in the next round's numbers.  Needs no GPU (without a device the selector assumes 256 CUs and
plans this library's own members for the plain dense pairs).  `python tests/test_plan_snapshot.py --write` regenerates."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bitblas_amd as bitblas  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plan_snapshot.json")
KEEP = ("name", "kernel_family", "threads", "grid", "split_k", "lds_bytes", "rows_per_wave", "batch_tile")


def configurations():
    f16 = dict(A_dtype="float16", out_dtype="float16", accum_dtype="float16")
    i8 = dict(A_dtype="int8", out_dtype="int32", accum_dtype="int32")
    f8 = dict(A_dtype="e4m3_float8", W_dtype="e4m3_float8", out_dtype="float16", accum_dtype="float32")
    out = []

    def add(tag, ms, strict=True, **kw):
        out.append((tag, ms, strict, kw))
    add("c1", [1], N=1024, K=1024, W_dtype="int4", **f16)
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)):       # c2 + the group-sized rows
        for strict in (True, False):
            add(f"c2_n{N}k{K}_{'strict' if strict else 'exact'}", [1, 2], strict, N=N, K=K, W_dtype="int4", group_size=128,
                with_scaling=True, **f16)
    add("c3", [3, 8, 16, 32, 64, 128, 256, 1024, 4096], N=4096, K=4096, W_dtype="uint4", group_size=128, with_scaling=True,
        with_zeros=True, zeros_mode="original", **f16)
    # prefill-sized M between the decode batches and the full chip: which ping-pong tile (256 / 128 rows) or the lockstep member
    add("c3_prefill", [512, 1024, 1536, 2048, 2560, 3072, 8192], N=4096, K=4096, W_dtype="uint4", group_size=128, with_scaling=True,
        with_zeros=True, zeros_mode="original", **f16)
    add("c3_prefill_n11008", [256, 512, 1024, 2048, 4096], N=11008, K=4096, W_dtype="uint4", group_size=128, with_scaling=True,
        with_zeros=True, zeros_mode="original", **f16)
    add("c4_prefill", [1024, 2048], N=4096, K=4096, W_dtype="int2", **i8)
    add("c5_prefill", [1024, 2048], N=8192, K=8192, **f8)
    add("c3_quantized_zeros", [16, 128, 4096], N=4096, K=4096, W_dtype="uint4", group_size=128, with_scaling=True,
        with_zeros=True, zeros_mode="quantized", **f16)
    add("c4", [1, 2, 16, 128, 4096], N=4096, K=4096, W_dtype="int2", **i8)
    for (name, N, K) in (("o", 8192, 8192), ("down", 8192, 28672), ("qkv", 10240, 8192), ("gate", 28672, 8192), ("gateup", 57344, 8192)):
        add(f"c5_{name}", [1, 16, 256, 4096], N=N, K=K, **f8)
        add(f"c5_{name}_shard8", [1, 4096], N=N // 8, K=K, **f8)
    for (N, K) in ((1024, 28672), (1280, 8192), (512, 11008)):                                       # per-rank shards: the K split
        add(f"shard_n{N}k{K}", [1, 2], N=N, K=K, W_dtype="int4", group_size=128, with_scaling=True, **f16)
    add("bf16_uint4", [1, 16, 4096], N=4096, K=4096, A_dtype="bfloat16", W_dtype="uint4", out_dtype="bfloat16", accum_dtype="float32",
        group_size=128, with_scaling=True)
    add("nf4", [1, 4096], N=4096, K=4096, W_dtype="nf4", group_size=128, with_scaling=True, **f16)
    add("int4_act", [1, 4096], N=4096, K=4096, A_dtype="int4", W_dtype="int2", out_dtype="int32", accum_dtype="int32")
    return out


def snapshot():
    keep_env = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("WQAA_")}      # plan-time switches off
    try:
        snap = {}
        for tag, ms, strict, kw in configurations():
            op = bitblas.Matmul(bitblas.MatmulConfig(M=ms, **kw), enable_tuning=False, strict_reference=strict)
            for m in ms:
                p = op.plans[m]
                snap[f"{tag}/m{m}"] = {k: p[k] for k in KEEP}
        # the group launches of the headline step
        for tag, ns in (("qkv", (4096, 4096, 4096)), ("gateup", (11008, 11008))):
            ops = [bitblas.Matmul(bitblas.MatmulConfig(M=1, N=n, K=4096, A_dtype="float16", W_dtype="int4", out_dtype="float16",
                                                       accum_dtype="float16", group_size=128, with_scaling=True),
                                  enable_tuning=False, strict_reference=False) for n in ns]
            g = bitblas.group_plan(ops, 1)
            snap[f"group_{tag}/m1"] = dict(launches=g["launches"], **{k: g["plan"][k] for k in KEEP})
            if tag == "gateup":          # gate_proj + up_proj + the gated activation as one launch (wqaa_matmul_gate_up)
                for m in (1, 2):
                    p = bitblas.gate_up_plan(bitblas.Matmul(bitblas.MatmulConfig(M=[1, 2], **{k: getattr(ops[0].config, k) for k in
                                             ("N", "K", "A_dtype", "W_dtype", "out_dtype", "accum_dtype", "group_size", "with_scaling")}),
                                             enable_tuning=False, strict_reference=False), m)
                    snap[f"pair_gateup/m{m}"] = {k: p[k] for k in KEEP}
                    pn = bitblas.gate_up_plan(ops[0], m, norm=True) if m == 1 else None      # ... with the RMSNorm in front
                    if pn is not None:
                        snap[f"pair_gateup_norm/m{m}"] = {k: pn[k] for k in KEEP}
        return snap
    finally:
        os.environ.update(keep_env)


def test_selector_answers_are_the_pinned_ones():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the snapshot is the device-less selection (256 CUs assumed, own members for the plain dense pairs)")
    with open(GOLDEN) as f:
        want = json.load(f)
    got = snapshot()
    assert sorted(got) == sorted(want)
    diff = {k: (want[k], got[k]) for k in want if want[k] != got[k]}
    assert not diff, "selector moved (regenerate with --write if intended, and say why in DESIGN.md):\n" + json.dumps(diff, indent=1)


if __name__ == "__main__":
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump(snapshot(), f, indent=1, sort_keys=True)
        print("wrote", GOLDEN)
    else:
        print(json.dumps(snapshot(), indent=1, sort_keys=True))
