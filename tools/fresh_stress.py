#!/usr/bin/env python
"""tools/fresh_stress.py [regex] [launches] [--large]: first launches on freshly uploaded operands, over the selector sweep's per-kernel examples.

For every (member class, mode, layout) example of tools/member_coverage.py whose class matches `regex` (default: the members with
hand-counted waits - ping-pong, decode and mid-M forms): build the operator and host operands once, take one launch as the reference,
then `launches` more, each on operands uploaded again from the host into new allocations, and compare the bits.  A counted wait that is
one piece short shows on such launches (cold translations, uneven first round trips) and never on warm repeats - how the 128-row
ping-pong tile's prologue was found in round 6 (profiles/r06_repetition.txt).  Prints one line per failing example and a summary."""
import os
import re
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bitblas_amd as bitblas  # noqa: E402
import member_coverage  # noqa: E402

TDT = {"float16": torch.float16, "bfloat16": torch.bfloat16, "e4m3_float8": torch.float8_e4m3fn, "e5m2_float8": torch.float8_e5m2}


def host_operands(ex, rng):
    M, N, K, a, w, mode = ex["M"], ex["N"], ex["K"], ex["a"], ex["w"], ex["mode"]
    op = bitblas.Matmul(bitblas.MatmulConfig(**ex["cfg"]), enable_tuning=False, strict_reference=ex["strict"])
    fp8 = ("e4m3_float8", "e5m2_float8")
    if a in ("float16", "bfloat16"):
        A = (torch.from_numpy(rng.random((M, K), dtype=np.float32)) - 0.5).to(TDT[a])
    elif a == "int8":
        A = torch.from_numpy(rng.integers(-128, 128, size=(M, K), dtype=np.int8))
    elif a == "int4":
        A = torch.from_numpy(rng.integers(-128, 128, size=(M, K // 2), dtype=np.int8))
    else:
        A = (torch.from_numpy(rng.random((M, K), dtype=np.float32)) * 2 - 1).to(TDT[a])
    native = w == a or (a in fp8 and w in fp8)
    scale = zeros = None
    if native:
        if a in ("float16", "bfloat16"):
            W = (torch.from_numpy(rng.random((N, K), dtype=np.float32)) - 0.5).to(TDT[a])
        elif a == "int8":
            W = torch.from_numpy(rng.integers(-128, 128, size=(N, K), dtype=np.int8))
        elif a == "int4":
            W = torch.from_numpy(rng.integers(-128, 128, size=(N, K // 2), dtype=np.int8))
        else:
            W = (torch.from_numpy(rng.random((N, K), dtype=np.float32)) * 2 - 1).to(TDT[w])
    else:
        src, bit = bitblas.Matmul.BITBLAS_TRICK_DTYPE_MAP[w]
        if src in ("fp_e4m3", "fp_e5m2"):
            W = (torch.from_numpy(rng.random((N, K), dtype=np.float32)) * 2 - 1).to(TDT[w]).view(torch.int8)
        else:
            W = torch.from_numpy(rng.integers(-128, 128, size=(N, K * bit // 8), dtype=np.int8))       # any bytes are a packed weight
        g = mode.get("group_size", -1)
        gg = K if g == -1 else g
        sdt = TDT.get(a, torch.float16)
        if mode.get("with_scaling"):
            scale = (torch.from_numpy(rng.random((N, K // gg), dtype=np.float32)) * 0.05).to(sdt)
        if mode.get("with_zeros"):
            zm = mode["zeros_mode"]
            if zm == "quantized":
                zeros = torch.from_numpy(rng.integers(-128, 128, size=(K // gg, N * bit // 8), dtype=np.int8))
            else:
                zt = torch.from_numpy(((1 << (bit - 1)) + rng.integers(-2, 3, size=(N, K // gg))).astype(np.float32)).to(sdt)
                zeros = (zt.float() * scale.float()).to(sdt) if zm == "rescale" else zt
    return op, A, W, scale, zeros


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = re.compile(argv[0] if len(argv) > 0 else r"pp|xdl|xmk")
    launches = int(argv[1]) if len(argv) > 1 else 20
    reach = member_coverage.reachable(with_args=True, per_kernel=True)
    small = {k: v for k, v in reach.items() if v["N"] * v["K"] <= (1 << 25) and v["M"] * v["N"] <= (1 << 25)}
    if "--large" in sys.argv:      # the classes only the large shapes reach (the 28672 x 8192-sized linears: K-split roundings, tail launches)
        covered = {k.split("|")[0] for k in small}
        reach = {c: v for c, v in member_coverage.reachable(with_args=True).items() if c not in covered}
        keys = sorted(k for k in reach if pat.search(k))
    else:
        reach = small
        keys = sorted(k for k in reach if pat.search(k.split("|")[0]))
    bad_keys, total = [], 0
    for k in keys:
        ex = reach[k]
        rng = np.random.default_rng(zlib.crc32(k.encode()))
        try:
            op, A, W, S, Z = host_operands(ex, rng)
            dev = lambda t: None if t is None else t.cuda()          # noqa: E731
            ref = op(dev(A), dev(W), scale=dev(S), zeros=dev(Z)).clone()
        except Exception as e:  # noqa: BLE001 - an example this generator cannot feed (reported, not counted)
            print(f"skip {k}: {type(e).__name__}: {str(e)[:100]}")
            continue
        torch.cuda.synchronize()
        keep, bad = [], 0
        for it in range(launches):
            pad = torch.empty(((it * 37) % 61 + 1) << 16, dtype=torch.uint8, device="cuda")
            t = (dev(A), dev(W), dev(S), dev(Z))
            out = op(t[0], t[1], scale=t[2], zeros=t[3])
            if not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
                bad += 1
            keep.append((t, out, pad))
        torch.cuda.synchronize()
        total += launches
        if bad:
            bad_keys.append((k, bad))
            print(f"FAIL {k}: {bad} of {launches} first launches differ (M = {ex['M']}, N = {ex['N']}, K = {ex['K']})", flush=True)
        del keep
    print(f"{len(keys)} examples matching /{pat.pattern}/, {total} first launches on fresh operands, {len(bad_keys)} examples with differences")
    return 1 if bad_keys else 0


if __name__ == "__main__":
    sys.exit(main())
