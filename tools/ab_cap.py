#!/usr/bin/env python
"""tools/ab_cap.py - same-process A/B of the GEMV grid cap (single launches, M = 1): default (grid capped at the
workgroups the chip holds at once, workgroups iterate over their XCD's eighth of the row-group blocks) against
WQAA_GEMV_UNCAP=1 (one row-group block per workgroup, the hardware dispatcher runs the surplus as slots free up).
hipGraph replays over rotating weight sets, microseconds per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    cases = [("int4", 22016, 4096, False), ("int4", 28672, 4096, False), ("int4", 28672, 8192, False), ("int4", 8192, 28672, False),
             ("int4", 28672, 8192, True), ("int4", 22016, 4096, True), ("fp8", 8192, 8192, True), ("fp8", 28672, 8192, True),
             ("fp8", 8192, 28672, True)]
    for (kind, N, K, strict) in cases:
        if kind == "int4":
            op = bench.get_op(1, N, K, strict=strict)
            nset = max(3, (640 << 20) // (N * K // 2))
            sets = [bench.make_linear(N, K, dev, gen)[1:3] for _ in range(nset)]
            A = (torch.rand((1, K), device=dev, generator=gen) - 0.5).half()
            nbytes = bench.algorithmic_bytes(1, N, K)
        else:
            op = bench.get_op(1, N, K, W_dtype="e4m3_float8", A_dtype="e4m3_float8", out_dtype="float16", scaling=False, accum="float32")
            nset = max(3, (640 << 20) // (N * K))
            sets = [((torch.rand((N, K), device=dev, generator=gen) * 2 - 1).to(torch.float8_e4m3fn), None) for _ in range(nset)]
            A = (torch.rand((1, K), device=dev, generator=gen) * 2 - 1).to(torch.float8_e4m3fn)
            nbytes = N * K + K + 2 * N
        out = torch.empty((1, N), dtype=torch.float16, device=dev)

        def launch_all():
            st = torch.cuda.current_stream(dev).cuda_stream
            for (w, sc) in sets:
                op.lib.run(A.data_ptr(), w.data_ptr(), None, sc.data_ptr() if sc is not None else None, None, None, out.data_ptr(), 1, st)

        row = []
        for uncap in ("0", "1", "0", "1"):
            os.environ["WQAA_GEMV_UNCAP"] = uncap
            plan = op.lib.plan(1)          # wqaa_select: bumps the plan epoch, the variable is re-read
            t = bench.graph_time(dev, launch_all, nset, replays=7)
            row.append(f"uncap={uncap} grid {plan['grid']:5d}: {t * 1e6:7.2f} us {nbytes / t / 1e9:6.0f} GB/s")
        os.environ["WQAA_GEMV_UNCAP"] = "0"
        op.lib.plan(1)
        print(f"{kind} {'strict' if strict else 'exact '} {N}x{K} [{plan['name']}]  " + " | ".join(row))


if __name__ == "__main__":
    main()
