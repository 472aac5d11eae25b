"""Golden vectors for the decoder layer's elementwise ops that ride in the GEMV launches (include/wqaa.h: WQAA_EPI_RMSNORM_INPUT,
wqaa_matmul_gate_up, WQAA_EPI_ADD_RESIDUAL), produced by RUNNING the reference's own code:

    BitnetRMSNorm.forward                     integration/BitNet/modeling_bitnet.py:89-104 - the class is cut out of the file where it
                                              lies (ast) and executed as it stands; the rest of that module needs the un-importable
                                              bitblas / flash-attn stack
    act_fn(gate) * up                         :240-244 / :281-287 with act_fn = ACT2FN[config.hidden_act], hidden_act = "silu"
                                              (configuration_bitnet.py; transformers' own ACT2FN table is imported)
    residual + hidden_states                  :854, :860

on seeded float16 inputs (torch CPU; a float16 op there is the fp32 op rounded once, as on the GPU).  Output:
tests/golden/layer_ops_golden.npz (committed).  Runs only where /root/reference exists.  Test infrastructure - never imported by
the product."""
from __future__ import annotations

import ast
import os
import sys

import numpy as np

REF_FILE = "/root/reference/integration/BitNet/modeling_bitnet.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "layer_ops_golden.npz")


def reference_rmsnorm_class():
    import torch
    from torch import nn
    src = open(REF_FILE).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BitnetRMSNorm")
    ns = {"torch": torch, "nn": nn}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF_FILE, "exec"), ns)
    return ns["BitnetRMSNorm"]


def main():
    if not os.path.exists(REF_FILE):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    import torch
    from transformers.activations import ACT2FN
    RMSNorm = reference_rmsnorm_class()
    act_fn = ACT2FN["silu"]
    gen = torch.Generator().manual_seed(20250925)
    out = {}
    for tag, (rows, K, eps, xmul) in {"a": (1, 4096, 1e-5, 1.0), "b": (2, 8192, 1e-6, 6.0), "c": (3, 1024, 1e-5, 0.02), "d": (1, 11008, 1e-6, 3.0)}.items():
        x = (torch.randn((rows, K), generator=gen) * xmul).half()
        norm = RMSNorm(K, eps=eps)
        with torch.no_grad():
            norm.weight.copy_(1.0 + (torch.rand(K, generator=gen) - 0.5) * 0.5)
        norm = norm.half()
        with torch.no_grad():
            y = norm(x)
        out[f"norm_{tag}_x"], out[f"norm_{tag}_w"], out[f"norm_{tag}_eps"], out[f"norm_{tag}_y"] = x.numpy(), norm.weight.detach().numpy(), np.float32(eps), y.numpy()
        gate = (torch.randn((rows, K), generator=gen) * 3.0).half()
        up = torch.randn((rows, K), generator=gen).half()
        out[f"act_{tag}_gate"], out[f"act_{tag}_up"], out[f"act_{tag}_y"] = gate.numpy(), up.numpy(), (act_fn(gate) * up).numpy()
        res = (torch.randn((rows, K), generator=gen) * 2.0).half()
        hid = torch.randn((rows, K), generator=gen).half()
        out[f"add_{tag}_residual"], out[f"add_{tag}_hidden"], out[f"add_{tag}_y"] = res.numpy(), hid.numpy(), (res + hid).numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
