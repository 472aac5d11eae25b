"""CPU oracle for the WqAa matmul hot path.  TEST INFRASTRUCTURE ONLY.

This module is the *checker*: a numpy restatement of what microsoft/BitBLAS defines for
`bitblas.Matmul` with a quantised weight operand.  Nothing under `bitblas_amd/` may import
it; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

What it restates (paths relative to the reference checkout):

* weight packing ........ `bitblas/quantization/utils.py:54-70`  (general_compress)
* LOP3 interleave ....... `bitblas/quantization/utils.py:73-110` (numpy) and
                          `bitblas/ops/lop3_permutate/lop3_permutate_impl.py:12-132` (TIR; the
                          version `Matmul.transform_weight` really runs - it differs from the numpy
                          helper for 1-bit/float16, where the numpy helper forgets its byte swizzle)
* per-element decoders .. `bitblas/quantization/quantization.py:141-156` (fp4), `:169-176` (e4m3
                          bit trick), `:185-230` (uint / int / int1 / uint-with-zeros)
* compute graph ......... `bitblas/ops/general_matmul/tirscript/matmul_dequantize_impl.py:339-499`
                          (decode -> scale/zeros in A_dtype -> sum over k in accum dtype -> cast to
                          out_dtype -> + bias), dense variant `tirscript/matmul_impl.py:50-86`
* NF4 table ............. `bitblas/ops/general_matmul/__init__.py:413-434`
* code offset ........... `bitblas/ops/general_matmul/__init__.py:685-696` (`clamp(W)+2^(b-1)`)
* the formulation the reference's own tests compare against (dequantise to half, matmul in
  fp32, cast, add bias): `testing/python/operators/test_general_matmul_ops_backend_tl.py:227-273`
* GPTQ unpack helpers ... `bitblas/module/__init__.py:24-74`

PARITY PINNING STATUS
---------------------
The reference cannot be imported here (its tvm / tilelang submodules are empty) and it has no CPU
matmul.  What *is* importable is `bitblas/quantization/utils.py`; `oracle/gen_golden.py` runs
`general_compress` / `interleave_weight` from that file and commits the vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks this module against them bit for bit.
=> packing + interleave: PINNED by reference-generated fixtures (4b and 2b for both targets, 1b for int8; the
   1b/float16 interleave follows the TIR op, which neither numpy copy of the helper reproduces - see above).
=> decode + matmul semantics: PINNED for the configurations the reference's own operator tests assert on.
   `oracle/gen_optest_golden.py` RUNS those test functions from the files where they lie (a recorder stands in
   for the un-importable `bitblas` package) and commits their seeded operands + the expected result of their
   in-test `ref_program` (tests/golden/optest_golden.*): 13 cases of test_general_matmul_ops_backend_tl.py
   (uint4 / int4, g = -1 / 32, zeros original / rescale / quantized, M = 1 / 256), 9 of
   test_general_matmul_ops_backend.py (bias, M = 1 / 768), 4 of test_general_matmul_fp8.py
   (W e4m3 x A fp16, +-scale g = 32), 2 of test_general_matmul_ops_nf4.py, 4 of test_general_matmul_bf16.py.
   `tests/test_optest_golden.py`: this module is bit-identical to those expectations on >= 99.8 % of the fp16
   outputs and one fp16 ulp away on the rest (fp32 summation order inside torch.matmul); the bf16 expectations
   are themselves rounded to bfloat16 and are met within 2^-7.
   Dense fp8 x fp8 (e4m3, e5m2): the reference test (test_general_matmul_fp8.py:11-71) PRINTS its expectation and
   asserts nothing; the printed tensor is recorded by the same script and `matmul_dense` reproduces it to 1e-5.
   The BitNet caller ops and the GPTQ repack are pinned by gen_bitnet_golden.py / gen_gptq_golden.py (the W_int2 x
   A_int8 path is thereby pinned at the layer level: quantisers, exact integer product, post-process).
=> STILL UNPINNED ("parity unpinned"): the operator-level W_int2 x A_int8 decode convention
   (u - 2, TE spec) and fp4_e2m1 - the reference never asserts on them - and the kernels' e4m3 bit trick (zero -> 2^-7, wrong subnormals): the fp8
   test's expectation decodes per IEEE and hides the difference behind rtol = 1e-1, so `strict_reference=True`
   for e4m3 follows `quantization.py:169-176` as read, not as run.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# dtype tables  (reference: Matmul.BITBLAS_TRICK_DTYPE_MAP, general_matmul/__init__.py:324-345)
# --------------------------------------------------------------------------------------
W_DTYPE_MAP = {
    "float64": ("fp", 64), "float32": ("fp", 32), "float16": ("fp", 16), "bfloat16": ("bf", 16),
    "int32": ("int", 32), "uint32": ("uint", 32), "int16": ("int", 16), "uint16": ("uint", 16),
    "int8": ("int", 8), "uint8": ("uint", 8), "int4": ("int", 4), "uint4": ("uint", 4),
    "int2": ("int", 2), "uint2": ("uint", 2), "int1": ("int", 1), "uint1": ("uint", 1),
    "nf4": ("nf", 4), "fp4_e2m1": ("fp", 4),
    "e4m3_float8": ("fp_e4m3", 8), "e5m2_float8": ("fp_e5m2", 8),
}

# reference: general_matmul/__init__.py:413-434
NF4_LUT = np.array([
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=np.float64)


# --------------------------------------------------------------------------------------
# packing
# --------------------------------------------------------------------------------------
def general_compress(codes: np.ndarray, source_bits: int = 4, storage_dtype=np.int8) -> np.ndarray:
    """Little-field-first bit packing along the last axis.

    out[..., j] = OR_k (codes[..., j*e + k] << (bits*k)),  e = 8 // bits
    (reference: quantization/utils.py:54-70, TIR twin quant_compress_impl.py:22-31).
    """
    e = 8 // source_bits
    c = np.asarray(codes)
    if c.dtype == np.float16:
        c = c.astype(np.int8)
    assert c.shape[-1] % e == 0
    c = c.astype(np.int64).reshape(*c.shape[:-1], c.shape[-1] // e, e)
    out = np.zeros(c.shape[:-1], dtype=np.int64)
    for k in range(e):
        # the reference ORs int8 values shifted in int8 arithmetic: bits above 8 fall off
        out |= (c[..., k] << (source_bits * k)) & 0xFF
    return out.astype(np.uint8).view(np.int8).view(storage_dtype)


def general_decompress(packed: np.ndarray, source_bits: int = 4) -> np.ndarray:
    """Inverse of general_compress: unsigned field values, last axis expanded by 8//bits."""
    e = 8 // source_bits
    p = np.asarray(packed).view(np.uint8).astype(np.int32)
    mask = (1 << source_bits) - 1
    fields = [(p >> (source_bits * k)) & mask for k in range(e)]
    return np.stack(fields, axis=-1).reshape(*p.shape[:-1], p.shape[-1] * e).astype(np.int8)


def interleave_field_positions(nbits: int, target_bits: int):
    """For one 32-bit word: dst bit offset of every source field (before the byte swizzles).

    field `o` (source bits [o*nbits, (o+1)*nbits)) moves to bit
        (o % G) * S + (o // G) * nbits,   S = target bits (16 | 8),  G = 32 // S
    (reference: quantization/utils.py:80-88).
    """
    S = target_bits
    G = 32 // S
    n = 32 // nbits
    return [(o % G) * S + (o // G) * nbits for o in range(n)]


def interleave_weight(qweight: np.ndarray, nbits: int = 4, target_dtype: str = "float16",
                      follow: str = "tir") -> np.ndarray:
    """LOP3 interleave of packed weights, 32 bits at a time.

    follow="tir"   -> what `LOP3Permutate` (the op `transform_weight` runs) computes
                      (lop3_permutate_impl.py:27-132): byte swizzles for f16/2b, f16/1b, int8/1b.
    follow="numpy" -> `quantization/utils.py:73-110` verbatim behaviour, including its missing
                      `return` for nbits=1/float16 (the swizzled value is computed and dropped).
    """
    assert target_dtype in ("float16", "int8", "int4")   # int4: the tir op only (lop3_permutate_impl.py:131-134)
    S = {"int8": 8, "int4": 4}.get(target_dtype, 16)
    q = np.ascontiguousarray(qweight).view(np.uint32).astype(np.uint64)
    out = np.zeros_like(q)
    mask = (1 << nbits) - 1
    for o, shift in enumerate(interleave_field_positions(nbits, S)):
        out |= ((q >> (nbits * o)) & mask) << shift
    out &= 0xFFFFFFFF

    def mv(x, m, right, left):
        return ((x & m) >> right) << left

    if nbits == 1 and target_dtype == "int8":
        r = out & 0xF0F00F0F
        r |= mv(out, 0x000000F0, 4, 16)
        r |= mv(out, 0x0000F000, 12, 24)
        r |= mv(out, 0x000F0000, 16, 4)
        r |= mv(out, 0x0F000000, 24, 12)
        out = r
    elif nbits == 2 and target_dtype == "float16":
        r = out & 0xFF0000FF
        r |= mv(out, 0x0000FF00, 8, 16)
        r |= mv(out, 0x00FF0000, 16, 8)
        out = r
    elif nbits == 1 and target_dtype == "float16" and follow == "tir":
        r = out & 0xF000000F
        r |= mv(out, 0x000000F0, 4, 8)
        r |= mv(out, 0x00000F00, 8, 16)
        r |= mv(out, 0x0000F000, 12, 24)
        r |= mv(out, 0x000F0000, 16, 4)
        r |= mv(out, 0x00F00000, 20, 12)
        r |= mv(out, 0x0F000000, 24, 20)
        out = r
    return (out & 0xFFFFFFFF).astype(np.uint32).view(np.int8).reshape(np.asarray(qweight).shape)


def deinterleave_weight(qweight: np.ndarray, nbits: int, target_dtype: str) -> np.ndarray:
    """Inverse of interleave_weight(follow="tir") - used to read reference-layout checkpoints."""
    probe = np.zeros(32, dtype=np.uint32)
    for b in range(32):
        probe[b] = np.uint32(1) << np.uint32(b)
    moved = interleave_weight(probe.view(np.int8), nbits, target_dtype).view(np.uint32)
    dst_of_src = [int(np.log2(int(v))) for v in moved]
    q = np.ascontiguousarray(qweight).view(np.uint32).astype(np.uint64)
    out = np.zeros_like(q)
    for src, dst in enumerate(dst_of_src):
        out |= ((q >> dst) & 1) << src
    return out.astype(np.uint32).view(np.int8).reshape(np.asarray(qweight).shape)


# --------------------------------------------------------------------------------------
# decoders: storage fields -> exact values (returned as float64 holding fp16-representable numbers,
# or int64 for integer activations)
# --------------------------------------------------------------------------------------
def decode_fp4(codes: np.ndarray) -> np.ndarray:
    """`fp4_e2m1` as the reference decodes it: 1 sign bit + 3 exponent bits, no mantissa.

    s = f4 >> 3; e = f4 & 7; e == 0 -> 0 else (-1)^s * 2^(e + 8 - 15)   (quantization.py:141-156)
    """
    c = np.asarray(codes).astype(np.int64) & 0xF
    s = c >> 3
    e = c & 7
    val = np.ldexp(1.0, (e | 8) - 15)
    val = np.where(s == 1, -val, val)
    return np.where(e == 0, 0.0, val)


def _as_u8(x) -> np.ndarray:
    x = np.asarray(x)
    if x.dtype.itemsize == 1:
        return x.view(np.uint8)
    return (x.astype(np.int64) & 0xFF).astype(np.uint8)


def decode_e4m3_strict(u8: np.ndarray) -> np.ndarray:
    """e4m3 byte -> fp16 with the reference's bit trick (quantization.py:169-176).

    f16 bits = s<<15 | (((v & 63) << 7) | (e4 << 8) | (e4 << 7)) ^ 0x2000,  e4 = v & 0x40.
    Exact for normal numbers; zero maps to 2^-7, subnormals / NaN are wrong by construction.
    """
    v = _as_u8(u8).astype(np.uint32)
    s = (v >> 7) << 15
    e4 = v & 0x40
    e = (((v & 63) << 7) | (e4 << 8) | (e4 << 7)) ^ 0x2000
    bits = ((s | e) & 0xFFFF).astype(np.uint16)
    return bits.view(np.float16).astype(np.float64)


def decode_e4m3_ieee(u8: np.ndarray) -> np.ndarray:
    """OCP e4m3fn byte -> exact value (what torch.float8_e4m3fn and gfx950 MFMA mean)."""
    v = _as_u8(u8).astype(np.int64)
    s = np.where((v >> 7) == 1, -1.0, 1.0)
    e = (v >> 3) & 0xF
    m = v & 7
    normal = np.ldexp(1.0 + m / 8.0, e - 7)
    sub = np.ldexp(m / 8.0, -6)
    val = np.where(e == 0, sub, normal)
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return s * val


def decode_e5m2(u8: np.ndarray) -> np.ndarray:
    """e5m2 byte -> value: it is the top byte of an IEEE half (quantization.py:179-182)."""
    v = _as_u8(u8).astype(np.uint16) << 8
    return v.view(np.float16).astype(np.float64)


def decode_codes(codes: np.ndarray, source_format: str, bit: int, strict_reference: bool = True,
                 lut: np.ndarray | None = None) -> np.ndarray:
    """Unsigned storage fields (N, K) -> decoded value before scale/zeros (float64, exact).

    uint: u;  int (bit>1): u - 2^(bit-1);  int1: sign-extend -> {0,-1};  nf: LUT[u];
    fp (fp4_e2m1), fp_e4m3, fp_e5m2: see the helpers above.
    (matmul_dequantize_impl.py:391-433; quantization.py:185-230)
    """
    c = np.asarray(codes)
    u = c.astype(np.int64) & ((1 << bit) - 1)
    if source_format == "uint":
        if bit == 8 and strict_reference:
            # 8-bit weights are not unpacked: the TE graph reads `B[n, k].astype(in_dtype)` from the int8 STORAGE buffer
            # (matmul_dequantize_impl.py:404-406, storage_dtype "int8"), so a "uint8" byte >= 128 decodes as a NEGATIVE
            # value.  Pinned by executing the definition (tests/golden/te_golden.npz, f16_uint8_scale); the reference's
            # own tests never notice because they draw uint8 weights below 128.  strict_reference=False: true unsigned
            return _as_u8(c).view(np.int8).astype(np.float64)
        return u.astype(np.float64)
    if source_format == "int":
        if bit == 8:
            return np.asarray(codes).astype(np.int8).astype(np.float64)
        if bit == 1:
            return np.where(u == 1, -1.0, 0.0)  # sign-extended 1-bit field
        return (u - (1 << (bit - 1))).astype(np.float64)
    if source_format == "nf":
        table = NF4_LUT if lut is None else np.asarray(lut, dtype=np.float64)
        return table[u]
    if source_format == "fp":
        return decode_fp4(u)
    if source_format == "fp_e4m3":
        return decode_e4m3_strict(u) if strict_reference else decode_e4m3_ieee(u)
    if source_format == "fp_e5m2":
        return decode_e5m2(u)
    raise ValueError(source_format)


def f16(x: np.ndarray) -> np.ndarray:
    return np.asarray(x).astype(np.float16)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what torch.bfloat16 arithmetic does
    after every operation; the TE definition materialises B_decode in A_dtype)."""
    u = np.ascontiguousarray(np.asarray(x, dtype=np.float32)).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)




def _quantized_zero_difference(codes, qzeros, bit: int, gi) -> np.ndarray:
    """`(w_u - zero_u)` of zeros_mode="quantized", taken in the int8 storage type as the TE expression
    does (quantization.py:197-217 with storage_dtype int8, matmul_dequantize_impl.py:375-389): exact
    for sub-byte fields, wraps mod 256 to a signed byte for 8-bit ones."""
    mask = (1 << bit) - 1
    zq = general_decompress(np.asarray(qzeros), bit).astype(np.int64) & mask   # (K/g, N) unsigned fields
    u = np.asarray(codes).astype(np.int64) & mask
    d = u - zq[gi, :].T
    return ((d + 128) % 256) - 128


# Large matrices (the Llama-70B linears: 235 M elements) are decoded in blocks of weight rows on a thread pool: the same
# function on every block (a row of B_decode depends on its own codes, Scale and Zeros only), concatenated - numpy's
# elementwise loops are single-threaded and release the GIL, and whole-output checks at M = 4096 need the decode in seconds.
_BLOCKED_ABOVE = 1 << 24
_BLOCK_ROWS = 512


def _blocked_rows(fn, codes, bit, scale, zeros, zeros_mode, kw):
    import concurrent.futures as cf
    import os
    N = codes.shape[0]
    per_byte = max(1, 8 // bit)
    quantized = zeros is not None and zeros_mode == "quantized"
    zeros_a = None if zeros is None else np.asarray(zeros)
    scale_a = None if scale is None else np.asarray(scale)

    def one(n0):
        n1 = min(N, n0 + _BLOCK_ROWS)
        z = None
        if zeros_a is not None:
            z = zeros_a[:, n0 // per_byte:(n1 + per_byte - 1) // per_byte] if quantized else zeros_a[n0:n1]
        return fn(codes[n0:n1], scale=None if scale_a is None else scale_a[n0:n1], zeros=z, **kw)

    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        parts = list(pool.map(one, range(0, N, _BLOCK_ROWS)))
    return np.concatenate(parts, axis=0)


def dequantize_weight_exact(codes: np.ndarray, source_format: str, bit: int, *, K: int | None = None,
                            scale=None, zeros=None, zeros_mode: str = "original", group_size: int = -1, lut=None) -> np.ndarray:
    """The real-valued dequantised weight, float64, NO rounding to A_dtype between the steps: what the
    `strict_reference=False` exact-product members compute with (csrc/wqaa_gemvx_kernel.h).  NOT the reference's
    definition - that is `dequantize_weight`, which rounds every step to A_dtype - but the value both approximate."""
    codes = np.asarray(codes)
    N, Kc = codes.shape
    K = Kc if K is None else K
    if N * Kc > _BLOCKED_ABOVE and N >= 2 * _BLOCK_ROWS:
        return _blocked_rows(dequantize_weight_exact, codes, bit, scale, zeros, zeros_mode,
                             dict(source_format=source_format, bit=bit, K=K, zeros_mode=zeros_mode, group_size=group_size, lut=lut))
    g = K if group_size in (-1, None) else group_size
    gi = np.arange(K) // g
    if zeros is not None and zeros_mode == "quantized":
        w = _quantized_zero_difference(codes, zeros, bit, gi).astype(np.float64)
    else:
        w = decode_codes(codes, source_format, bit, False, lut).astype(np.float64)
    if scale is None:
        return w
    s = np.asarray(scale).astype(np.float64)[:, gi]
    if zeros is None or zeros_mode == "quantized":
        return w * s
    z = np.asarray(zeros).astype(np.float64)[:, gi]
    if zeros_mode == "original":
        return (w - z) * s
    if zeros_mode == "rescale":
        return w * s - z
    raise ValueError(zeros_mode)


def matmul_dequant_exact(A: np.ndarray, codes: np.ndarray, *, source_format: str, bit: int, scale=None, zeros=None,
                         zeros_mode="original", group_size=-1, bias=None, out_dtype="float16", lut=None) -> np.ndarray:
    """C = cast_out(sum_k A[m,k] * w[n,k]) (+ bias after the cast) with the unrounded w of dequantize_weight_exact."""
    A = np.asarray(A)
    K = A.shape[-1]
    Wd = dequantize_weight_exact(codes, source_format, bit, K=K, scale=scale, zeros=zeros, zeros_mode=zeros_mode,
                                 group_size=group_size, lut=lut)
    acc = _gemm_nt(A.reshape(-1, K), Wd, np.float64)
    out = _cast_out(acc, out_dtype)
    if bias is not None:
        out = (out + np.asarray(bias).astype(out.dtype)).astype(out.dtype)
    return out.reshape(*A.shape[:-1], Wd.shape[0])


def dequantize_weight(codes: np.ndarray, source_format: str, bit: int, *, K: int | None = None,
                      scale=None, zeros=None, zeros_mode: str = "original", group_size: int = -1,
                      a_dtype: str = "float16", strict_reference: bool = True, lut=None) -> np.ndarray:
    """B_decode of the TE spec, materialised in A_dtype (matmul_dequantize_impl.py:391-451).

    codes: (N, K) unsigned storage fields (i.e. what general_compress packed).
    scale: (N, K/g) in A_dtype.  zeros: (N, K/g) A_dtype for original/rescale, or the *packed*
    (K/g, N*bit/8) int8 array for "quantized".
    Every arithmetic step rounds to A_dtype exactly as the TE expression does:
      original : (w - z) * s        rescale: w * s - z        quantized: (w_u - z_u) * s
    """
    codes = np.asarray(codes)
    N, Kc = codes.shape
    K = Kc if K is None else K
    if N * Kc > _BLOCKED_ABOVE and N >= 2 * _BLOCK_ROWS:
        return _blocked_rows(dequantize_weight, codes, bit, scale, zeros, zeros_mode,
                             dict(source_format=source_format, bit=bit, K=K, zeros_mode=zeros_mode, group_size=group_size,
                                  a_dtype=a_dtype, strict_reference=strict_reference, lut=lut))
    g = K if group_size in (-1, None) else group_size
    gi = np.arange(K) // g
    if a_dtype == "int8":
        assert scale is None and zeros is None
        return decode_codes(codes, source_format, bit, strict_reference, lut).astype(np.int64)
    if a_dtype == "bfloat16":
        return _dequantize_weight_bf16(codes, source_format, bit, K, scale, zeros, zeros_mode, g, gi, lut)
    ft = {"float16": np.float16, "float32": np.float32}[a_dtype]
    with_zeros = zeros is not None
    if with_zeros and zeros_mode == "quantized":
        w = _quantized_zero_difference(codes, zeros, bit, gi).astype(ft)
    else:
        w = decode_codes(codes, source_format, bit, strict_reference, lut).astype(ft)
    if scale is None:
        return w
    s = np.asarray(scale).astype(ft)[:, gi]
    if not with_zeros:
        return (w * s).astype(ft)
    if zeros_mode == "original":
        z = np.asarray(zeros).astype(ft)[:, gi]
        return ((w - z).astype(ft) * s).astype(ft)
    if zeros_mode == "rescale":
        z = np.asarray(zeros).astype(ft)[:, gi]
        return ((w * s).astype(ft) - z).astype(ft)
    if zeros_mode == "quantized":
        return (w * s).astype(ft)
    raise ValueError(zeros_mode)


def _dequantize_weight_bf16(codes, source_format, bit, K, scale, zeros, zeros_mode, g, gi, lut):
    """bfloat16 flavour of dequantize_weight: every operation rounds to bf16 (values held as float32).
    Scale / Zeros arrive as float32 arrays holding bf16-representable values."""
    with_zeros = zeros is not None
    if with_zeros and zeros_mode == "quantized":
        w = bf16_round(_quantized_zero_difference(codes, zeros, bit, gi).astype(np.float32))
    else:
        # e4m3: exact IEEE decode - the reference's bit trick exists for float16 only
        # (quantization.py:169-176 asserts dtype == "float16")
        w = bf16_round(decode_codes(codes, source_format, bit, source_format != "fp_e4m3", lut).astype(np.float32))
    if scale is None:
        return w
    sc = bf16_round(np.asarray(scale, dtype=np.float32))[:, gi]
    if not with_zeros or zeros_mode == "quantized":
        return bf16_round(w * sc)
    z = bf16_round(np.asarray(zeros, dtype=np.float32))[:, gi]
    if zeros_mode == "original":
        return bf16_round(bf16_round(w - z) * sc)
    if zeros_mode == "rescale":
        return bf16_round(bf16_round(w * sc) - z)
    raise ValueError(zeros_mode)


_OUT_NP = {"float16": np.float16, "float32": np.float32, "int32": np.int32, "int8": np.int8,
           "bfloat16": np.float32}


def _gemm_nt(A: np.ndarray, W: np.ndarray, ct=None) -> np.ndarray:
    """A @ W^T in the floating type `ct` (default: the arrays' own).  torch's CPU kernels where torch is importable - its
    threaded BLAS takes a 4096^3 float64 product in well under a second where numpy's bundled one takes ~16 s, and its
    float16 -> float conversions are vectorised where numpy's are scalar loops - numpy otherwise: the same sums either way
    up to the order of the additions."""
    A, W = np.asarray(A), np.asarray(W)
    ct = np.dtype(ct or A.dtype)
    try:
        import torch
        tt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[ct]
        return torch.matmul(torch.from_numpy(np.ascontiguousarray(A)).to(tt), torch.from_numpy(np.ascontiguousarray(W)).to(tt).T).numpy()
    except ImportError:
        return A.astype(ct) @ W.astype(ct).T


def _cast_out(acc: np.ndarray, out_dtype: str) -> np.ndarray:
    """accumulator -> float32 -> out_dtype (the one cast of the TE graph), through torch's vectorised conversions when the
    array is large and torch is importable (same IEEE round-to-nearest-even conversions, bit for bit)."""
    if acc.size >= (1 << 22) and out_dtype in ("float16", "float32") and acc.dtype in (np.float64, np.float32):
        try:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(acc)).float()
            return (t.half() if out_dtype == "float16" else t).numpy()
        except ImportError:
            pass
    return acc.astype(np.float32).astype(_OUT_NP[out_dtype])


def exact_int_matmul(A: np.ndarray, W: np.ndarray) -> np.ndarray:
    """A @ W^T of integer arrays, exact, as int64.  numpy has no BLAS for integer dtypes (a 4096^3 product takes minutes in
    its loops); every partial sum here is an integer below K * max|a| * max|w|, and while that bound is below 2^53 a float64
    GEMM adds integers exactly whatever its summation order - so whole M = 4096 outputs can be checked, not sampled rows."""
    A = np.asarray(A)
    W = np.asarray(W)
    K = A.shape[-1]
    amax = int(np.abs(A.astype(np.int64)).max()) if A.size else 0
    wmax = int(np.abs(W.astype(np.int64)).max()) if W.size else 0
    if K * amax * wmax < (1 << 53):
        return np.rint(_gemm_nt(A, W, np.float64)).astype(np.int64)
    return A.astype(np.int64) @ W.astype(np.int64).T


def matmul_dequant(A: np.ndarray, codes: np.ndarray, *, source_format: str, bit: int,
                   scale=None, zeros=None, zeros_mode="original", group_size=-1, bias=None,
                   a_dtype="float16", out_dtype="float16", strict_reference=True, lut=None,
                   wide: bool = True) -> np.ndarray:
    """C = cast_out(sum_k A[m,k] * B_decode[n,k]) (+ bias after the cast).

    The sum is taken in float64 (wide=True) or float32; the reference leaves the accumulation
    order to its code generator, the tests compare against an fp32 matmul.  Integer activations
    accumulate exactly.
    """
    A = np.asarray(A)
    K = A.shape[-1]
    Wd = dequantize_weight(codes, source_format, bit, K=K, scale=scale, zeros=zeros,
                           zeros_mode=zeros_mode, group_size=group_size, a_dtype=a_dtype,
                           strict_reference=strict_reference, lut=lut)
    A2 = A.reshape(-1, K)
    if a_dtype == "int8":
        acc = exact_int_matmul(A2, Wd)
        out = acc.astype(_OUT_NP[out_dtype]) if out_dtype.startswith("int") else acc.astype(_OUT_NP[out_dtype])
    else:
        acc = _gemm_nt(A2, Wd, np.float64 if wide else np.float32)
        out = _cast_out(acc, out_dtype)
    if bias is not None:
        out = (out + np.asarray(bias).astype(out.dtype)).astype(out.dtype)
    return out.reshape(*A.shape[:-1], Wd.shape[0])


def matmul_dense(A: np.ndarray, W: np.ndarray, *, a_dtype: str, w_dtype: str | None = None,
                 out_dtype: str = "float16", bias=None) -> np.ndarray:
    """Dense nt matmul C = A @ W^T (tirscript/matmul_impl.py:50-86). fp8 operands come as bytes."""
    w_dtype = a_dtype if w_dtype is None else w_dtype

    def val(x, dt):
        if dt == "e4m3_float8":
            return decode_e4m3_ieee(x)
        if dt == "e5m2_float8":
            return decode_e5m2(x)
        return np.asarray(x)

    Av, Wv = val(A, a_dtype), val(W, w_dtype)
    K = Av.shape[-1]
    if a_dtype in ("int8", "uint8"):
        acc = exact_int_matmul(Av.reshape(-1, K), Wv)
        out = acc.astype(_OUT_NP[out_dtype])
    else:
        acc = _gemm_nt(Av.reshape(-1, K), Wv, np.float64)
        out = _cast_out(acc, out_dtype)
    if bias is not None:
        out = (out + np.asarray(bias).astype(out.dtype)).astype(out.dtype)
    return out.reshape(*Av.shape[:-1], Wv.shape[0])


# --------------------------------------------------------------------------------------
# weight preparation as Matmul.transform_weight does it (general_matmul/__init__.py:662-711)
# --------------------------------------------------------------------------------------
def unpack_int4_activations(A_packed: np.ndarray) -> np.ndarray:
    """(M, K/2) int8 -> (M, K) int64: two's-complement nibbles, low nibble = even k.  The packing of the
    reference's int4 test: `(A[:, ::2] & 0x0F) + ((A[:, 1::2] & 0x0F) << 4)`
    (testing/python/operators/test_general_matmul_ops_int4.py:49)."""
    u = _as_u8(A_packed).astype(np.int64)
    lo, hi = u & 0xF, (u >> 4) & 0xF
    out = np.empty((u.shape[0], u.shape[1] * 2), dtype=np.int64)
    out[:, 0::2] = lo
    out[:, 1::2] = hi
    return np.where(out >= 8, out - 16, out)


def matmul_int4_act(A_packed: np.ndarray, codes: np.ndarray, *, w_bits: int, out_dtype: str = "int32") -> np.ndarray:
    """W_int4 / W_int2 x A_int4 (BitNet a4.8 family), int32 accumulation.

    A: packed two's-complement nibbles.  `codes`: the (N, K) raw fields of the weight operand.
    4-bit fields are native two's-complement int4 (the operands of the int4 tensor-core instruction,
    tilelang/dequantize/matmul_dequantize_mma.py:512-520, dense pair ("int4", "int4") of
    general_matmul/__init__.py:41).  2-bit fields are ZERO-extended to a nibble, signed or not: both
    the plain decode (ibid. :742-749: `(x >> 0) & 3`, `(x >> 2) & 3`) and the fast decode
    (gpu/intrin/lop3.py:1057-1083, whose `isSigned` branch is empty) do exactly that.  The reference
    test compares with `A.float() @ B.T.float()` on non-negative operands
    (test_general_matmul_ops_int4.py:141-146), which this restates for the full nibble range."""
    A = unpack_int4_activations(A_packed)
    w = np.asarray(codes).astype(np.int64) & ((1 << w_bits) - 1)
    if w_bits == 4:
        w = np.where(w >= 8, w - 16, w)
    acc = A @ w.T
    return acc.astype({"int32": np.int32, "float32": np.float32}[out_dtype])


def weight_to_codes(weight: np.ndarray, source_format: str, bit: int) -> np.ndarray:
    """int formats (<8 bit): clamp(W, -2^(b-1), 2^(b-1)) + 2^(b-1), in int8 arithmetic; others: as int8."""
    w = np.asarray(weight)
    if source_format == "int" and bit < 8:
        maxq = 1 << (bit - 1)
        return (np.clip(w, -maxq, maxq).astype(np.int8) + np.int8(maxq)).astype(np.int8)
    return w.astype(np.int8) if w.dtype != np.int8 else w


def transform_weight(weight: np.ndarray, source_format: str, bit: int, *, fast_decoding: bool,
                     a_dtype: str = "float16") -> np.ndarray:
    """codes -> (N, K*bit/8) int8 bytes in the reference's checkpoint layout."""
    codes = weight_to_codes(weight, source_format, bit)
    if bit not in (1, 2, 4):
        return codes
    packed = general_compress(codes, bit)
    if fast_decoding:
        packed = interleave_weight(packed, bit, "int8" if a_dtype == "int8" else "float16")
    return packed


# --------------------------------------------------------------------------------------
# GPTQ helpers (module/__init__.py:24-74)
# --------------------------------------------------------------------------------------
def unpack_qzeros(qzeros: np.ndarray, bits: int, v2: bool = False) -> np.ndarray:
    q = np.ascontiguousarray(qzeros).view(np.int32)
    e = 32 // bits
    cols = [(q >> (bits * i)).astype(np.int8) for i in range(e)]
    un = np.stack(cols, axis=-1).reshape(q.shape[0], q.shape[1] * e)
    if not v2:
        un = (un + 1).astype(np.int8)
    return un & np.int8((1 << bits) - 1)


def unpack_qweight(qweight: np.ndarray, bits: int) -> np.ndarray:
    q = np.ascontiguousarray(qweight).view(np.int8)
    e = 8 // bits
    cols = [(q >> (bits * i)).astype(np.int8) for i in range(e)]
    un = np.stack(cols, axis=-1).reshape(q.shape[0], q.shape[1] * e)
    return un & np.int8((1 << bits) - 1)


# --------------------------------------------------------------------------------------
# BitNet caller ops (integration/BitNet/utils_quant.py:150-176, 205-216)
# --------------------------------------------------------------------------------------
def bitnet_weight_quant(weight: np.ndarray):
    """ternary weights: s = 1 / clamp(mean|W|, 1e-5); round(W * s) clamped to [-1, 1] (:150-155)."""
    w = np.asarray(weight, dtype=np.float32)
    s = np.float32(1.0) / np.maximum(np.abs(w).mean(dtype=np.float32), np.float32(1e-5))
    return np.clip(np.rint(w * s), -1, 1).astype(np.int8), np.float32(s)


def bitnet_activation_quant(x: np.ndarray):
    """per-token int8: s = Qp / clamp(max|x|, 1e-5); round(x * s) clamped to [-128, 127] (:162-169).

    `Qp / tensor` with a Python int on the left is `tensor.reciprocal() * Qp` in torch (Tensor.__rtruediv__):
    two fp32 roundings, not one quotient - pinned by tests/golden/bitnet_golden.npz (reference-run)."""
    xf = np.asarray(x).astype(np.float32)
    m = np.maximum(np.abs(xf).max(axis=-1, keepdims=True), np.float32(1e-5))
    s = ((np.float32(1.0) / m).astype(np.float32) * np.float32(127.0)).astype(np.float32)
    q = np.clip(np.rint(xf * s), -128, 127).astype(np.int8)
    return q, s.astype(np.float32)


def bitnet_forward(x: np.ndarray, weight_q: np.ndarray, sw, bias=None) -> np.ndarray:
    """`BitLinearBitBLAS.forward` (:205-216): quantise, exact int matmul -> float32, / si, / sw, half, + bias."""
    q, si = bitnet_activation_quant(x)
    acc = (q.astype(np.int64) @ np.asarray(weight_q).astype(np.int64).T).astype(np.float32)
    out = (acc / si).astype(np.float32)
    out = (out / np.float32(sw)).astype(np.float32).astype(np.float16)
    if bias is not None:
        out = (out + np.asarray(bias).astype(np.float16)).astype(np.float16)
    return out


# --------------------------------------------------------------------------------------
# tolerance helper with the reference's semantics (bitblas/testing/__init__.py:29-91)
# --------------------------------------------------------------------------------------
def silu_mul_f16(gate: np.ndarray, up: np.ndarray) -> np.ndarray:
    """The gated activation the reference's callers run in front of down_proj (integration/BitNet/modeling_bitnet.py:240-244,
    :281-287: `act_fn(gate) * up`, act_fn = silu) as torch evaluates it on float16 tensors: silu in fp32
    (x / (1 + exp(-x))) rounded to float16, then the float16 product (fp32 multiply, one rounding)."""
    g = np.asarray(gate, dtype=np.float16).astype(np.float32)
    act = f16(g / (np.float32(1.0) + np.exp(-g, dtype=np.float32)))
    return f16(act.astype(np.float32) * np.asarray(up, dtype=np.float16).astype(np.float32))


def rms_norm_f16(x: np.ndarray, weight: np.ndarray, eps: float) -> np.ndarray:
    """BitnetRMSNorm.forward (= LlamaRMSNorm; integration/BitNet/modeling_bitnet.py:99-104) on float16 tensors: the variance and
    the scaling in fp32, rounded to float16, then the float16 product with the weight."""
    h = np.asarray(x, dtype=np.float16).astype(np.float32)
    variance = np.mean(h * h, axis=-1, keepdims=True, dtype=np.float32)
    h = f16(h * (np.float32(1.0) / np.sqrt(variance + np.float32(eps), dtype=np.float32)))
    return f16(np.asarray(weight, dtype=np.float16).astype(np.float32) * h.astype(np.float32))


def add_residual_f16(out: np.ndarray, residual: np.ndarray) -> np.ndarray:
    """`residual + linear(x)` on float16 tensors (modeling_bitnet.py:839-860): fp32 add of the two float16 values, one rounding"""
    return f16(np.asarray(out, dtype=np.float16).astype(np.float32) + np.asarray(residual, dtype=np.float16).astype(np.float32))


def count_mismatch(a: np.ndarray, b: np.ndarray, rtol: float, atol: float) -> int:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return int((np.abs(a - b) > atol + rtol * np.abs(b)).sum())
