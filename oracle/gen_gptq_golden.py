"""Golden vectors for the GPTQ repack path, produced by RUNNING the reference's own code.

`bitblas/module/__init__.py` is plain torch around `bitblas.Matmul`; executed from where it lies with a stub
`bitblas` package (only `general_compress` is real - the reference's numpy helper, loaded standalone) this script
calls, on seeded AutoGPTQ-shaped tensors (qweight (K/8*bits, N) int32, scales (K/g, N) half, qzeros (K/g, N/8*bits)
int32):

    unpack_qweight / unpack_qzeros / unpack_qzeros_v2          (:24-74)
    Linear.repack_from_gptq / repack_from_gptq_v2              (:315-363) as unbound functions on a bare `self`
        whose `bitblas_matmul.weight_transform` is None (the unpacked integer codes stay visible) - for the three
        zeros modes

and records inputs and resulting buffers.  Output: tests/golden/gptq_golden.npz (committed).
Runs only where /root/reference exists.  Test infrastructure - never imported by the product.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/bitblas"
REF_FILE = os.path.join(REF, "module", "__init__.py")
REF_UTILS = os.path.join(REF, "quantization", "utils.py")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "gptq_golden.npz")


def load_reference_module():
    import torch
    spec = importlib.util.spec_from_file_location("ref_quant_utils", REF_UTILS)
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    bb = types.ModuleType("bitblas")
    cache = types.ModuleType("bitblas.cache")
    cache.global_operator_cache = object()
    cache.get_database_path = lambda: "/nonexistent"
    quant = types.ModuleType("bitblas.quantization")
    qutils = types.ModuleType("bitblas.quantization.utils")
    qutils.general_compress = utils.general_compress
    quant.utils = qutils
    bb.cache, bb.quantization = cache, quant
    bb.Matmul = bb.MatmulConfig = object
    bb.auto_detect_nvidia_target = lambda: "stub"
    sys.modules.update({"bitblas": bb, "bitblas.cache": cache, "bitblas.quantization": quant,
                        "bitblas.quantization.utils": qutils})
    torch.Tensor.cuda = lambda self, *a, **k: self
    ns = {"__name__": "ref_bitblas_module", "__file__": REF_FILE}
    exec(compile(open(REF_FILE).read(), REF_FILE, "exec"), ns)
    return ns


def pack_rows(fields: np.ndarray, bits: int) -> np.ndarray:
    """AutoGPTQ packing: consecutive ROWS of `fields` (R, C) into int32 words -> (R * bits / 32, C)."""
    e = 32 // bits
    out = np.zeros((fields.shape[0] // e, fields.shape[1]), dtype=np.uint32)
    for i in range(e):
        out |= fields[i::e].astype(np.uint32) << np.uint32(bits * i)
    return out.view(np.int32)


def pack_cols(fields: np.ndarray, bits: int) -> np.ndarray:
    e = 32 // bits
    out = np.zeros((fields.shape[0], fields.shape[1] // e), dtype=np.uint32)
    for i in range(e):
        out |= fields[:, i::e].astype(np.uint32) << np.uint32(bits * i)
    return out.view(np.int32)


def main():
    if not os.path.exists(REF_FILE):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    import torch
    ref = load_reference_module()
    Linear = ref["Linear"]
    rng = np.random.default_rng(20250925)
    out = {}
    for bits in (4, 2):
        N, K, g = 64, 256, 64
        codes = rng.integers(0, 1 << bits, size=(K, N))            # (in, out) as AutoGPTQ holds them
        zstored = rng.integers(0, 1 << bits, size=(K // g, N))     # stored zero points (full range: 2^bits - 1 wraps)
        qweight = torch.from_numpy(pack_rows(codes, bits))
        qzeros = torch.from_numpy(pack_cols(zstored, bits))
        scales = torch.from_numpy((rng.random((K // g, N), dtype=np.float32) * 0.1 + 0.01).astype(np.float16))
        tag = f"b{bits}"
        out.update({f"{tag}_qweight": qweight.numpy(), f"{tag}_qzeros": qzeros.numpy(), f"{tag}_scales": scales.numpy(),
                    f"{tag}_unpack_qweight": ref["unpack_qweight"](qweight.T.contiguous().view(torch.int8), bits).numpy(),
                    f"{tag}_unpack_qzeros": ref["unpack_qzeros"](qzeros, bits).numpy(),
                    f"{tag}_unpack_qzeros_v2": ref["unpack_qzeros_v2"](qzeros, bits).numpy()})
        gptq = types.SimpleNamespace(qweight=qweight, qzeros=qzeros, scales=scales, bias=None)
        for v2 in (False, True):
            for mode in ("original", "rescale", "quantized"):
                zbuf = (torch.zeros((K // g, N * bits // 8), dtype=torch.int8) if mode == "quantized"
                        else torch.zeros((N, K // g), dtype=torch.float16))
                self = types.SimpleNamespace(
                    TORCH_STORAGE_DTYPE=torch.int8, bits=bits, torch_dtype=torch.float16, bias=None, zeros=zbuf,
                    bitblas_matmul=types.SimpleNamespace(weight_transform=None,
                                                         config=types.SimpleNamespace(zeros_mode=mode)))
                if v2:
                    Linear.repack_from_gptq_v2(self, gptq)
                else:
                    Linear.repack_from_gptq(self, gptq, device="cpu")
                key = f"{tag}_{'v2' if v2 else 'v1'}_{mode}"
                # with weight_transform = None the reference leaves `qweight` = the transposed int8 VIEW of the
                # packed words; the integer codes it would hand to weight_transform are unpack_qweight of that view
                out[f"{key}_scales"] = self.scales.numpy()
                out[f"{key}_zeros"] = self.zeros.numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e3:.0f} kB)", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
