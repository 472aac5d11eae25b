"""Generate golden packing vectors by RUNNING the reference's own numpy helpers.

Runs only in the build container (needs /root/reference).  It loads
`/root/reference/bitblas/quantization/utils.py` standalone (that file depends on numpy/torch only)
and records, for seeded random inputs:

  general_compress(codes, bits)                       for bits in {1, 2, 4}
  interleave_weight(compressed, bits, target_dtype)   for target in {"float16", "int8"}

Outputs: tests/golden/packing_golden.npz  (committed; a few hundred KB).
Test infrastructure - never imported by the product.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

REF_UTILS = "/root/reference/bitblas/quantization/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "packing_golden.npz")


def load_reference_utils():
    spec = importlib.util.spec_from_file_location("ref_quant_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_test_suite_interleave():
    """`interleave_weight` as copied into the reference's own type-conversion test (numpy only); the file itself
    needs tvm, so the function's source text is sliced out at run time and executed - nothing is copied here."""
    path = "/root/reference/testing/python/type_conversion/test_int4b_fp16_convert.py"
    src = open(path).read()
    start = src.index("def interleave_weight(")
    end = src.index("def tir_interleave_weight(")
    ns = {"np": np}
    exec(compile(src[start:end], path, "exec"), ns)
    return ns["interleave_weight"]


def main():
    if not os.path.exists(REF_UTILS):
        print("reference not present; golden vectors are already committed", file=sys.stderr)
        return 0
    ref = load_reference_utils()
    test_copy = load_test_suite_interleave()
    rng = np.random.default_rng(20250704)
    out = {}
    skipped = set()
    for bits in (1, 2, 4):
        for (n, k) in ((4, 64), (16, 256), (3, 32)):
            codes = rng.integers(0, 1 << bits, size=(n, k), dtype=np.int8)
            comp = ref.general_compress(codes, source_bits=bits, storage_dtype=np.int8)
            tag = f"b{bits}_n{n}_k{k}"
            out[f"codes_{tag}"] = codes
            out[f"compress_{tag}"] = comp
            if (k * bits // 8) % 4 == 0:
                for tgt in ("float16", "int8"):
                    try:
                        inter = ref.interleave_weight(comp.copy(), nbits=bits, target_dtype=tgt)
                    except OverflowError as exc:
                        # numpy >= 2 rejects the helper's np.int32(0xF0F00F0F)-style masks (1b/int8,
                        # 2b/f16, 1b/f16 branches).  The reference's test suite carries its own copy of the
                        # helper with the masks wrapped in np.uint32 (testing/python/type_conversion/
                        # test_int4b_fp16_convert.py:29-65): run THAT for these branches.  1b/f16 is left out:
                        # both copies compute the byte swizzle and then return the unswizzled word, while the
                        # TIR op transform_weight really runs (lop3_permutate_impl.py:12-132) applies it.
                        skipped.add(f"{bits}b/{tgt}: {type(exc).__name__}")
                        if (bits, tgt) in ((1, "int8"), (2, "float16")):
                            inter = test_copy(comp.copy(), nbits=bits, target_dtype=tgt)
                            out[f"interleave_{tgt}_{tag}"] = np.asarray(inter).view(np.int8).reshape(comp.shape)
                        continue
                    out[f"interleave_{tgt}_{tag}"] = np.asarray(inter).view(np.int8).reshape(comp.shape)
    # signed sources as transform_weight feeds them (codes already offset to unsigned)
    w = rng.integers(-8, 8, size=(8, 128), dtype=np.int8)
    out["int4_signed_src"] = w
    out["int4_signed_compress"] = ref.general_compress((w + 8).astype(np.int8), source_bits=4,
                                                       storage_dtype=np.int8)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays; reference helper failed for: {sorted(skipped)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
