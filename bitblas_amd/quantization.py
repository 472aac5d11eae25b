"""Host-side packing helpers with the reference's names (bitblas/quantization/utils.py:54-110).

`general_compress` and `interleave_weight` are pure-numpy like upstream (they are called by user
code to build `zeros` for `zeros_mode="quantized"`, e.g. module/__init__.py:333-336); the operator's
own weight path uses the C packer in libwqaa_hip.so instead.
"""
from __future__ import annotations

import numpy as np


def general_compress(lowprecision_weight, source_bits=4, storage_dtype=np.int8):
    """Pack `8 // source_bits` fields per byte along the last axis, lowest field first."""
    per_byte = 8 // source_bits
    w = np.asarray(lowprecision_weight)
    if w.dtype == np.float16:
        w = w.astype(np.int8)
    lead, last = w.shape[:-1], w.shape[-1]
    grouped = w.reshape(*lead, last // per_byte, per_byte).astype(np.uint8)
    shifts = (np.arange(per_byte, dtype=np.uint8) * source_bits).astype(np.uint8)
    packed = np.bitwise_or.reduce(np.left_shift(grouped, shifts), axis=-1).astype(np.uint8)
    return packed.view(np.int8).view(storage_dtype)


def _swizzle(x, stay, moves):
    out = x & np.uint32(stay)
    for mask, right, left in moves:
        out |= ((x & np.uint32(mask)) >> np.uint32(right)) << np.uint32(left)
    return out


def interleave_weight(qweight, nbits=4, target_dtype="float16"):
    """LOP3 interleave, 32 bits at a time, as `LOP3Permutate` computes it
    (lop3_permutate_impl.py:27-132).  For nbits=1 / float16 this applies the nibble swizzle that the
    upstream numpy helper computes but forgets to return."""
    assert target_dtype in ("float16", "int8")
    stride = 8 if target_dtype == "int8" else 16
    groups = 32 // stride
    q = np.ascontiguousarray(qweight).view(np.uint32)
    out = np.zeros_like(q)
    mask = np.uint32((1 << nbits) - 1)
    for o in range(32 // nbits):
        shift = (o % groups) * stride + (o // groups) * nbits
        out |= ((q >> np.uint32(nbits * o)) & mask) << np.uint32(shift)
    if nbits == 1 and target_dtype == "int8":
        out = _swizzle(out, 0xF0F00F0F, [(0x000000F0, 4, 16), (0x0000F000, 12, 24),
                                         (0x000F0000, 16, 4), (0x0F000000, 24, 12)])
    elif nbits == 2 and target_dtype == "float16":
        out = _swizzle(out, 0xFF0000FF, [(0x0000FF00, 8, 16), (0x00FF0000, 16, 8)])
    elif nbits == 1 and target_dtype == "float16":
        out = _swizzle(out, 0xF000000F, [(0x000000F0, 4, 8), (0x00000F00, 8, 16), (0x0000F000, 12, 24),
                                         (0x000F0000, 16, 4), (0x00F00000, 20, 12), (0x0F000000, 24, 20)])
    return out.view(np.int8).reshape(np.asarray(qweight).shape)


def gen_quant4(k, n, groupsize=-1):
    """Synthetic 4-bit symmetric quantisation of a random `(k, n)` half matrix, the helper the reference's GPTQ / QuantLinear
    tests draw their weights from (bitblas/quantization/utils.py:8-51; used by testing/python/module/test_repack_from_gptq*.py
    and integration/pytorch/test_bitblas_quant_linear.py).  Groups of `groupsize` consecutive k share, per output column,
    the scale `absmax / 8`; codes are `clamp(round(w / s) + 8, 0, 16)` (the upper bound 16 is upstream's, kept).

    Returns `(original_w (k, n) half, nn.Linear(k, n) holding the dequantised weight, s (k / groupsize, n) half,
    signed codes (k, n) int32)`; consumes the torch CPU generator exactly as upstream does (one `randn((k, n), half)`)."""
    import torch
    import torch.nn as nn
    levels = 16
    original_w = torch.randn((k, n), dtype=torch.half, device="cpu")
    g = k if groupsize == -1 else groupsize
    grouped = original_w.view(k // g, g, n)
    s = grouped.abs().amax(dim=1, keepdim=True) * (2 / levels)          # exact in half: a power of two
    codes = torch.clamp(torch.round(grouped / s).int() + levels // 2, 0, levels)
    signed = codes - levels // 2
    dequant = signed.half() * s
    linear = nn.Linear(k, n, bias=False)
    linear.weight.data = dequant.reshape(k, n).t()
    return original_w, linear, s.reshape(k // g, n).contiguous(), signed.reshape(k, n).contiguous()
