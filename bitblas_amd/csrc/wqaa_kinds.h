// wqaa_kinds.h - what the GEMV and GEMM kernel families share: mode/flag enums, per-kind layout
// traits, the decode context and the output epilogue.
#pragma once
#include "wqaa_common.h"
#include "wqaa_decode.h"

namespace wqaa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// activation operand type of the kernel.  AT_I4: packed int4 activations - the int8 machine path with the
// nibbles widened to int8 while the tile is staged into LDS; only the LOP3 target width differs (4)
enum : int { AT_F16 = 0, AT_I8 = 1, AT_F8 = 2, AT_I4 = 3 };
constexpr bool at_is_int(int at) { return at == AT_I8 || at == AT_I4; }
// dequant arithmetic (matmul_dequantize_impl.py:435-449)
enum : int { MD_NONE = 0, MD_S = 1, MD_ZO = 2, MD_ZR = 3, MD_ZQ = 4 };
enum : int { FL_STRICT = 1, FL_A8 = 2, FL_ABF8 = 4, FL_BF16 = 8, FL_AQ = 16 };   // FL_AQ: int8 GEMV quantises fp16 activations itself   // FL_BF16: the 16-bit float type is bfloat16  // e4m3 reference bit trick; activations stored as fp8 (GEMV); fp8 MFMA activations are e5m2

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Which row-group blocks a GEMV workgroup works on: first, first + stride, ... < end.
// XCD-aware (workgroup b runs on XCD b % 8): every XCD owns a CONTIGUOUS eighth of the blocks - the 2-byte results that
// share a 128-byte line of C are written through one L2 - and its gridDim.x / 8 workgroups stride over that eighth, so
// a grid smaller than the block count (the chip holds only so many workgroups) or larger than it (a short member of a
// group launch) still loads the eight XCDs evenly.  With gridDim.x = the block count rounded up to 8 this is the plain
// swizzle blk = (b & 7) * (gridDim.x >> 3) + (b >> 3), one block per workgroup.
struct RowBlocks {
  int first, stride, end;
};
__host__ __device__ __forceinline__ RowBlocks xcd_row_blocks(int b, int grid, int n_blocks) {
  if (grid & 7) return RowBlocks{b, grid, n_blocks};
  const int chunk = (n_blocks + 7) >> 3;
  const int lo = (b & 7) * chunk;
  const int hi = lo + chunk < n_blocks ? lo + chunk : n_blocks;
  return RowBlocks{lo + (b >> 3), grid >> 3, hi};
}

template <int KIND, int AT>
struct KindTraits {
  static constexpr int BITS = (KIND == DK_INT4 || KIND == DK_LUT4) ? 4
                              : (KIND == DK_INT2)                  ? 2
                              : (KIND == DK_INT1)                  ? 1
                              : (KIND == DK_NATIVE && AT == AT_F16) ? 16
                                                                    : 8;
  static constexpr int EPW = 32 / BITS;          // elements per 32-bit word
  static constexpr int E = 128 / BITS;           // elements per 16-byte lane chunk
  static constexpr int PE = AT == AT_F16 ? 8 : 16;  // activation elements per 16-byte LDS piece
  static constexpr int G = cmax(EPW, PE);        // decode unit (elements)
  static constexpr int WPU = G / EPW;            // words per unit
  static constexpr int PU = G / PE;              // LDS pieces per unit
  static constexpr int UNITS = E / G;            // units per lane chunk
  static constexpr int PIECES = E / PE;          // LDS pieces per lane chunk
  static constexpr int S = AT == AT_F16 ? 16 : AT == AT_I4 ? 4 : 8;   // LOP3 interleave target width
  static constexpr bool SUBBYTE = BITS < 8;

  static constexpr int field_of_slot(int xs) {
    if (!SUBBYTE) return xs;
    if (at_is_int(AT)) return I8Unpack<BITS < 8 ? BITS : 4>::field_of_slot(xs);
    if (KIND == DK_LUT4) return lut_field_of_slot(xs);
    return F16Unpack<BITS < 8 ? BITS : 4>::field_of_slot(xs);
  }
  // source element (inside the unit) that lands in extraction slot x
  static constexpr int src_elem(int layout, int x) {
    const int wi = x / EPW, xs = x % EPW;
    if (!SUBBYTE) return x;
    return wi * EPW + src_of_field(BITS, S, layout, field_of_slot(xs));
  }
};

// ------------------------------------------------------------------------------------------
// epilogue: cast to out_dtype, then + bias in out_dtype (the TE graph adds Bias after the cast,
// matmul_dequantize_impl.py:462-477)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_round(float x) {
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return __builtin_bit_cast(float, u & 0xFFFF0000u);
}

__device__ __forceinline__ void store_out(void* C, long idx, float acc, int out_dtype, bool has_bias,
                                          float bias) {
  switch (out_dtype) {
    case WQAA_F16: {
      half_t v = (half_t)acc;
      if (has_bias) v = v + (half_t)bias;
      reinterpret_cast<half_t*>(C)[idx] = v;
    } break;
    case WQAA_F32: {
      float v = acc;
      if (has_bias) v = v + bias;
      reinterpret_cast<float*>(C)[idx] = v;
    } break;
    case WQAA_BF16: {
      float v = bf16_round(acc);
      if (has_bias) v = bf16_round(v + bias);
      reinterpret_cast<uint16_t*>(C)[idx] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
    } break;
    default: break;
  }
}
__device__ __forceinline__ void store_out(void* C, long idx, int acc, int out_dtype, bool has_bias,
                                          int bias) {
  switch (out_dtype) {
    case WQAA_I32: reinterpret_cast<int*>(C)[idx] = acc + (has_bias ? bias : 0); break;
    case WQAA_I8: reinterpret_cast<int8_t*>(C)[idx] = (int8_t)((int8_t)acc + (has_bias ? (int8_t)bias : 0)); break;
    case WQAA_F32: reinterpret_cast<float*>(C)[idx] = (float)acc + (has_bias ? (float)bias : 0.f); break;
    case WQAA_F16: {
      half_t v = (half_t)(float)acc;
      if (has_bias) v = v + (half_t)(float)bias;
      reinterpret_cast<half_t*>(C)[idx] = v;
    } break;
    default: break;
  }
}

// caller epilogue of the int8 path (utils_quant.py:170-176): out = input / si; out = out / sw; half; + bias
// IEEE division by a divisor many quotients share (a row's activation scale, the tensor's weight scale).  The compiler's own
// fp32 `a / b` IS correctly rounded here (v_div_scale x 2, v_rcp, the Newton step, two quotient corrections, v_div_fmas,
// v_div_fixup: 11 operations, checked in the ISA) - and when neither operand needs scaling (v_div_scale passes them
// through and clears VCC) and none is special, that sequence is exactly: r = rcp(b) refined once; q = a r; two rounds of
// e = fma(-b, q, a), q = fma(e, r, q).  `safe()` states the operand range in which that holds for quotients taken one after
// the other (a an integer's float or 0, divisors in [2^-30, 2^30]: every exponent difference stays under 96, nothing comes
// near a denormal); outside it the callers take `a / b`.  Bit for bit the same result, 5 operations per quotient.
struct ExactDiv {
  float b, r;
  __device__ __forceinline__ static ExactDiv prepare(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return ExactDiv{b, r};
  }
  __device__ __forceinline__ static bool safe(float b) {
    const float m = __builtin_fabsf(b);
    return m >= 0x1p-30f && m <= 0x1p30f;          // (false for NaN)
  }
  __device__ __forceinline__ float operator()(float a) const {
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
  }
  // two quotients at a time: the same five operations as packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth per issue)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ f32x2 operator()(f32x2 a) const {
    const f32x2 nb = {-b, -b}, rr = {r, r};
    f32x2 q = a * rr;
    f32x2 e = __builtin_elementwise_fma(nb, q, a);
    q = __builtin_elementwise_fma(e, rr, q);
    e = __builtin_elementwise_fma(nb, q, a);
    return __builtin_elementwise_fma(e, rr, q);
  }
};

__device__ __forceinline__ void store_out_fused(void* C, long idx, int acc, float row_scale, float tensor_scale,
                                                bool has_bias, const void* bias, int n) {
  // two IEEE fp32 divisions, like torch's `out / si / sw` (integration/BitNet/utils_quant.py:205-216)
  float v = (float)acc / row_scale;
  v = v / tensor_scale;
  half_t h = (half_t)v;
  if (has_bias) h = h + reinterpret_cast<const half_t*>(bias)[n];
  reinterpret_cast<half_t*>(C)[idx] = h;
}

// 8 two's-complement nibbles (element 2i = low nibble of byte i) -> 8 int8 in element order
__device__ __forceinline__ void widen_nibbles(uint32_t w, uint32_t& lo4, uint32_t& hi4) {
  const uint32_t even = w & 0x0F0F0F0Fu, odd = (w >> 4) & 0x0F0F0F0Fu;      // elements 0,2,4,6 / 1,3,5,7
  const uint32_t a = __builtin_amdgcn_perm(odd, even, 0x05010400u);           // e0 e1 e2 e3
  const uint32_t b = __builtin_amdgcn_perm(odd, even, 0x07030602u);           // e4 e5 e6 e7
  lo4 = sub_bytes(a ^ 0x08080808u, 0x08080808u);                             // (n ^ 8) - 8: sign extension
  hi4 = sub_bytes(b ^ 0x08080808u, 0x08080808u);
}

__device__ __forceinline__ half_t bits_to_half(uint32_t b) { return __builtin_bit_cast(half_t, (uint16_t)(b & 0xFFFFu)); }

// ---- bfloat16 flavour of the 16-bit float path (A_dtype = bfloat16, test_general_matmul_bf16.py) ----
// Integer fields are exact in bf16; `w * Scale` is one fp32 multiply rounded to bf16 by
// v_cvt_pk_bf16_f32 (round to nearest even) - the TE definition's single rounding in A_dtype.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x2_t as_bf2(uint32_t u) { return __builtin_bit_cast(bf16x2_t, u); }
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// two bfloat16 values (one register) times a float, rounded back to bfloat16
__device__ __forceinline__ uint32_t bf16x2_scale(uint32_t bits, float sc) {
  const float lo = __builtin_bit_cast(float, bits << 16), hi = __builtin_bit_cast(float, bits & 0xFFFF0000u);
  return cvt_pk_bf16(lo * sc, hi * sc);
}

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __builtin_bit_cast(float, (b & 0xFFFFu) << 16); }

// fields of one 32-bit word -> EPW/2 packed bf16 pairs.  PAIR_ORDER selects which two fields share a
// register: 0 = F16Unpack's extraction order (field i, field i + EPW/2) used by the GEMV's permuted LDS
// tile, 1 = natural order (2i, 2i+1) used by the MFMA fragments.  Plain layout only (bf16 has no LOP3).
// One more bfloat16 rounding between the two steps of the zero-point modes, as the TE expression has it
// (matmul_dequantize_impl.py:441-444 in bfloat16 arithmetic): original (w - z) * s, rescale w * s - z.
__device__ __forceinline__ void bf16x2_round(float& a, float& b) {
  const uint32_t r = cvt_pk_bf16(a, b);
  a = __builtin_bit_cast(float, r << 16);
  b = __builtin_bit_cast(float, r & 0xFFFF0000u);
}
// one dequantised value in bfloat16 arithmetic: ZM 0 = (w - zf) * s with an integer zf (w - zf exact: one rounding),
// 1 = original: bf16(bf16(w - z) * s), 2 = rescale: bf16(bf16(w * s) - z)
template <int ZM>
__device__ __forceinline__ void dequant_pair_bf16(float& a, float& b, float zf, float s, float z, bool scale) {
  if constexpr (ZM == 1) {
    a -= z; b -= z;
    bf16x2_round(a, b);
    a *= s; b *= s;
  } else if constexpr (ZM == 2) {
    a *= s; b *= s;
    bf16x2_round(a, b);
    a -= z; b -= z;
  } else {
    a -= zf; b -= zf;
    if (scale) { a *= s; b *= s; }
  }
}
template <int BITS, int PAIR_ORDER, int ZM = 0>
__device__ __forceinline__ void unpack_word_bf16(uint32_t w, float zf, float s, bool scale, uint32_t (&out)[32 / BITS / 2], float z = 0.f) {
  constexpr int EPW = 32 / BITS, NPAIR = EPW / 2;
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    const int f0 = PAIR_ORDER ? 2 * i : i, f1 = PAIR_ORDER ? 2 * i + 1 : i + NPAIR;
    float a = (float)__builtin_amdgcn_ubfe(w, f0 * BITS, BITS);
    float b = (float)__builtin_amdgcn_ubfe(w, f1 * BITS, BITS);
    dequant_pair_bf16<ZM>(a, b, zf, s, z, scale);
    out[i] = cvt_pk_bf16(a, b);
  }
}

// ---- runtime part of the decode: signedness is a data value, never a branch ----
struct DecodeCtx {
  half_t zf;        // folded integer zero point (signed formats: 2^(bits-1))
  uint32_t flip;    // int1 signed: ~w ; int8 signed: w ^ 0x80808080
  half_t off8;      // int8 weights: 1024 (+128 signed)
  uint32_t magic[8];  // F16Unpack magic exponent words, pinned in VGPRs (make_magic)
};


}  // namespace wqaa
