"""Golden vectors for the BitNet caller ops, produced by RUNNING the reference's own functions.

`integration/BitNet/utils_quant.py` is plain torch around a `bitblas.Matmul`; with a stub `bitblas` package
(nothing in it is exercised) and `torch.compile` disabled (the decorated methods then run as the eager functions
they wrap) the file executes from where it lies and this script calls

    BitLinearBitBLAS.weight_quant(W)                         (:155-160, static)
    BitLinearBitBLAS.activation_quant(self, x)               (:162-169; `self` is unused)
    BitLinearBitBLAS.post_quant_process(self, acc, si, sw)   (:171-176; `self` is unused)
    sw = 1 / W.abs().mean().clamp(min=1e-5)                  (:144 / :199, restated inline - one torch expression)

on seeded inputs; the int8 x ternary product between them is an exact integer matmul (numpy int64), cast to
float32 as the operator's out_dtype (:62).  Output: tests/golden/bitnet_golden.npz (committed).
Runs only where /root/reference exists.  Test infrastructure - never imported by the product.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF_FILE = "/root/reference/integration/BitNet/utils_quant.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bitnet_golden.npz")


def load_reference_module():
    os.environ["TORCH_COMPILE_DISABLE"] = "1"
    import torch
    import torch._dynamo
    torch._dynamo.config.disable = True
    bb = types.ModuleType("bitblas")
    cache = types.ModuleType("bitblas.cache")
    cache.global_operator_cache = object()
    cache.get_database_path = lambda: "/nonexistent"
    bb.cache = cache
    bb.Matmul = bb.MatmulConfig = object
    bb.auto_detect_nvidia_target = lambda: "stub"
    sys.modules["bitblas"], sys.modules["bitblas.cache"] = bb, cache
    ns = {"__name__": "ref_bitnet_utils_quant", "__file__": REF_FILE}
    exec(compile(open(REF_FILE).read(), REF_FILE, "exec"), ns)
    return ns


def main():
    if not os.path.exists(REF_FILE):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    import torch
    ref = load_reference_module()
    cls = ref["BitLinearBitBLAS"]
    out = {}
    gen = torch.Generator().manual_seed(20250924)
    for tag, (rows, N, K, wstd, xmul, bias) in {
            "a": (1, 64, 512, 0.02, 1.0, False), "b": (7, 96, 1024, 0.05, 3.0, True), "c": (33, 32, 256, 1.0, 0.01, False)}.items():
        W = torch.randn((N, K), generator=gen) * wstd
        x = (torch.randn((rows, K), generator=gen) * xmul).half()
        x[0, :8] = 0
        if tag == "c":
            x[1] = 0          # an all-zero token: the clamp(min=1e-5) branch
        wq = cls.weight_quant(W)
        sw = 1 / W.abs().mean().clamp(min=1e-5)
        q, si = cls.activation_quant(None, x)
        acc = torch.from_numpy((q.numpy().astype(np.int64) @ wq.numpy().astype(np.int64).T).astype(np.float32))
        y = cls.post_quant_process(None, acc, si, sw)
        b = None
        if bias:
            b = torch.randn((N,), generator=gen).half()
            y = y + b.view(1, -1).expand_as(y)      # forward(), :213-215
        assert wq.dtype == torch.int8 and q.dtype == torch.int8 and y.dtype == torch.float16
        out.update({f"{tag}_W": W.numpy(), f"{tag}_x": x.numpy(), f"{tag}_wq": wq.numpy(), f"{tag}_sw": sw.numpy(),
                    f"{tag}_q": q.numpy(), f"{tag}_si": si.numpy(), f"{tag}_y": y.numpy()})
        if b is not None:
            out[f"{tag}_bias"] = b.numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e3:.0f} kB)", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
