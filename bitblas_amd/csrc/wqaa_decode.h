// wqaa_decode.h - sub-byte weight unpack for gfx950, in registers.
//
// Native counterpart of the reference's CUDA "LOP3" decode snippets
// (bitblas/gpu/intrin/lop3.py:14-1097) and of the per-element TIR decoders
// (bitblas/quantization/quantization.py:141-230).  The numerics follow the TIR/TE definition:
// int formats subtract 2^(bits-1) (not the older "-7" convention of the C++ test header).
//
// Idea: a 32-bit word of packed weights is expanded with one V_AND_OR_B32 per *pair* of fields
// straight into a packed-half register: OR-ing a field into the mantissa of a suitably chosen
// fp16 "magic" exponent yields (2^(10-b) + q) exactly; one V_PK_ADD_F16 removes the offset (and
// folds an integer zero point).  Fields come out in "extraction order" x = 0..EPW-1; which source
// element k that is depends on the storage layout (plain general_compress order or the reference's
// LOP3-interleaved checkpoint order) and is described by the constexpr tables below.  Kernels do
// not permute weights: they permute the (tiny) activation tile once while staging it into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wqaa {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

enum : int { LAYOUT_PLAIN = 0, LAYOUT_LOP3 = 1 };

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ half2_t splat(half_t v) { return half2_t{v, v}; }

// ---------------------------------------------------------------------------------------------
// Layout algebra (compile time).
//   field f   : bits [f*BITS, (f+1)*BITS) of the stored 32-bit word
//   source o  : element index k % EPW inside the word, as general_compress numbered it
// LOP3 interleave (lop3_permutate_impl.py:27-35): source o -> bit (o % G) * S + (o / G) * BITS with
// S = 16 (float16 target) or 8 (int8 target), G = 32 / S, followed by the byte / nibble swizzles of
// the f16/2b, f16/1b and int8/1b variants (:37-132).
// ---------------------------------------------------------------------------------------------
constexpr int lop3_nibble_move(int bits, int S, int nib) {
  // returns the destination nibble of nibble `nib` under the variant's swizzle
  if (bits == 1 && S == 8) {
    // 0xF0F00F0F stay; n1->n4, n3->n6, n4->n1, n6->n3
    constexpr int mv[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    return mv[nib];
  }
  if (bits == 2 && S == 16) {
    // bytes 1 <-> 2
    constexpr int mv[8] = {0, 1, 4, 5, 2, 3, 6, 7};
    return mv[nib];
  }
  if (bits == 1 && S == 16) {
    // n1->n2, n2->n4, n3->n6, n4->n1, n5->n3, n6->n5
    constexpr int mv[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    return mv[nib];
  }
  return nib;
}

constexpr int lop3_dst_bit(int bits, int S, int o) {
  const int G = 32 / S;
  const int b = (o % G) * S + (o / G) * bits;
  return lop3_nibble_move(bits, S, b / 4) * 4 + (b % 4);
}

// source element stored in field f
constexpr int src_of_field(int bits, int S, int layout, int f) {
  if (layout == LAYOUT_PLAIN) return f;
  const int n = 32 / bits;
  for (int o = 0; o < n; ++o)
    if (lop3_dst_bit(bits, S, o) == f * bits) return o;
  return -1;
}

// ---------------------------------------------------------------------------------------------
// fp16 target: word -> EPW/2 packed-half registers.
// extraction slot x = 2*i + h  (i = pair index, h = half) holds field  f = i + (EPW/2) * h,
// because pair i takes the field at bit offset BITS*i of the low 16 bits and its twin in the
// high 16 bits.
// ---------------------------------------------------------------------------------------------
template <int BITS>
struct F16Unpack {
  static constexpr int EPW = 32 / BITS;   // elements per 32-bit word
  static constexpr int NPAIR = EPW / 2;
  static constexpr int field_of_slot(int x) { return (x >> 1) + NPAIR * (x & 1); }

  // offset to remove for the pair at in-half bit position b (b < 8 after the >>8)
  static __device__ __forceinline__ half_t magic_value(int b) { return (half_t)(float)(1 << (10 - b)); }
  static constexpr uint32_t magic_bits(int b) { return (uint32_t)((25 - b) << 10) * 0x00010001u; }

  // out[i] = (q_lo - zf, q_hi - zf), zf integer valued (exact).
  // `magic[b]` must hold magic_bits(b) in VGPRs (make_magic below): V_AND_OR_B32 takes one literal
  // (gfx9 constant bus), so the mask stays a literal and the magic exponent word is a register the
  // compiler cannot fold back - otherwise it splits the operation into v_and + v_or.
  static __device__ __forceinline__ void run(uint32_t w, half_t zf, const uint32_t (&magic)[8], half2_t (&out)[NPAIR]) {
    constexpr uint32_t fmask = (1u << BITS) - 1u;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int bit = BITS * i;           // position inside the 16-bit half
      const int b = bit & 7;              // position after the optional >> 8
      const uint32_t src = (bit >= 8) ? (w >> 8) : w;
      const uint32_t m = (fmask << b) * 0x00010001u;
      const uint32_t t = (src & m) | magic[b];          // V_AND_OR_B32
      const half_t off = (half_t)((float)(1 << (10 - b))) + zf;
      out[i] = as_h2(t) - splat(off);                  // V_PK_ADD_F16 (exact)
    }
  }
};

// the eight magic exponent words, pinned in VGPRs once per kernel
__device__ __forceinline__ void make_magic(uint32_t (&magic)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    magic[b] = (uint32_t)((25 - b) << 10) * 0x00010001u;
    asm volatile("" : "+v"(magic[b]));
  }
}

// 8-bit integer weights -> half: byte | 0x6400 == 1024 + u   (u8), signed via u ^ 0x80
template <bool SIGNED>
__device__ __forceinline__ void unpack8_f16(uint32_t w, half_t zf, half2_t (&out)[2]) {
  if (SIGNED) w ^= 0x80808080u;
  const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u);  // {b0,0x64,b1,0x64}
  const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u);  // {b2,0x64,b3,0x64}
  const half_t off = (half_t)(SIGNED ? 1152.0f : 1024.0f) + zf;
  out[0] = as_h2(lo) - splat(off);
  out[1] = as_h2(hi) - splat(off);
}

// ---------------------------------------------------------------------------------------------
// 16-entry half LUT (nf4, and the reference's sign+3-bit-exponent "fp4_e2m1") via V_PERM_B32.
// The table lives in 8 SGPR/VGPR words: lo[j] = low bytes of entries 4j..4j+3, hi[j] = high bytes.
// extraction: slots 0..3 = fields 0,2,4,6 ; slots 4..7 = fields 1,3,5,7
// ---------------------------------------------------------------------------------------------
struct Lut16 {
  uint32_t lo[4];
  uint32_t hi[4];
};

__device__ __forceinline__ Lut16 make_lut16(const half_t* tbl) {
  Lut16 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t l = 0, h = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint16_t bits = __builtin_bit_cast(uint16_t, tbl[4 * j + e]);
      l |= (uint32_t)(bits & 0xFF) << (8 * e);
      h |= (uint32_t)(bits >> 8) << (8 * e);
    }
    t.lo[j] = l;
    t.hi[j] = h;
  }
  return t;
}

constexpr int lut_field_of_slot(int x) { return x < 4 ? 2 * x : 2 * (x - 4) + 1; }

__device__ __forceinline__ void lut16_quad(const Lut16& t, uint32_t idx4, half2_t& o0, half2_t& o1) {
  const uint32_t sel = idx4 & 0x07070707u;
  const uint32_t m = ((idx4 >> 3) & 0x01010101u) * 0xFFu;
  const uint32_t l0 = __builtin_amdgcn_perm(t.lo[1], t.lo[0], sel);
  const uint32_t l1 = __builtin_amdgcn_perm(t.lo[3], t.lo[2], sel);
  const uint32_t h0 = __builtin_amdgcn_perm(t.hi[1], t.hi[0], sel);
  const uint32_t h1 = __builtin_amdgcn_perm(t.hi[3], t.hi[2], sel);
  const uint32_t l = (l1 & m) | (l0 & ~m);
  const uint32_t h = (h1 & m) | (h0 & ~m);
  o0 = as_h2(__builtin_amdgcn_perm(h, l, 0x05010400u));  // {l0,h0,l1,h1}
  o1 = as_h2(__builtin_amdgcn_perm(h, l, 0x07030602u));  // {l2,h2,l3,h3}
}

__device__ __forceinline__ void lut16_word(const Lut16& t, uint32_t w, half2_t (&out)[4]) {
  lut16_quad(t, w & 0x0F0F0F0Fu, out[0], out[1]);
  lut16_quad(t, (w >> 4) & 0x0F0F0F0Fu, out[2], out[3]);
}

// the reference's fp4 decode table: s = q>>3, e = q&7, e==0 -> 0 else (-1)^s 2^(e-7)
// (quantization.py:141-156)
__device__ __forceinline__ Lut16 make_fp4_lut(bool bf16 = false) {
  half_t tbl[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = q & 7;
    // half: sign | (e + 8) << 10 (= 2^(e-7)); bfloat16: sign | (e - 7 + 127) << 7 - both exact
    const uint16_t hbits = e == 0 ? (uint16_t)0 : (uint16_t)((((q >> 3) << 5) | (e | 8)) << 10);
    const uint16_t bbits = e == 0 ? (uint16_t)0 : (uint16_t)(((q >> 3) << 15) | ((e + 120) << 7));
    tbl[q] = __builtin_bit_cast(half_t, bf16 ? bbits : hbits);
  }
  return make_lut16(tbl);
}

// ---------------------------------------------------------------------------------------------
// fp8 weights -> half (A_dtype float16).  One word = 4 bytes -> 2 packed halves, natural order.
//   strict : the reference's bit trick, exact for normals, 0 -> 2^-7 (quantization.py:169-176)
//   ieee   : OCP e4m3fn value: ((v & 0x7f) << 7) as half, times 2^8, sign restored (NaN 0x7f -> 480
//            is not reproduced; NaN weights are outside the contract)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread_bytes_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0C010C00u); }
__device__ __forceinline__ uint32_t spread_bytes_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0C030C02u); }

__device__ __forceinline__ half2_t e4m3_pair_strict(uint32_t x /* bytes in bits 0-7 and 16-23 */) {
  const uint32_t s = (x & 0x00800080u) << 8;
  const uint32_t e4 = x & 0x00400040u;
  uint32_t r = ((x & 0x003F003Fu) << 7) | (e4 << 8) | (e4 << 7);
  r ^= 0x20002000u;
  return as_h2(r | s);
}
__device__ __forceinline__ half2_t e4m3_pair_ieee(uint32_t x) {
  const uint32_t s = (x & 0x00800080u) << 8;
  const half2_t mag = as_h2((x & 0x007F007Fu) << 7) * splat((half_t)256.0f);
  return as_h2(as_u32(mag) | s);
}
template <bool STRICT>
__device__ __forceinline__ void unpack_e4m3_f16(uint32_t w, half2_t (&out)[2]) {
  const uint32_t a = spread_bytes_lo(w), b = spread_bytes_hi(w);
  out[0] = STRICT ? e4m3_pair_strict(a) : e4m3_pair_ieee(a);
  out[1] = STRICT ? e4m3_pair_strict(b) : e4m3_pair_ieee(b);
}
__device__ __forceinline__ void unpack_e5m2_f16(uint32_t w, half2_t (&out)[2]) {
  out[0] = as_h2(__builtin_amdgcn_perm(0u, w, 0x010C000Cu));  // {0,b0,0,b1}
  out[1] = as_h2(__builtin_amdgcn_perm(0u, w, 0x030C020Cu));  // {0,b2,0,b3}
}

// ---------------------------------------------------------------------------------------------
// int8 target (A int8): word -> EPW/4 registers of four int8 each.
// slot x = 4*j + c (j = extraction step, c = byte lane) holds field f = c * (8/BITS) + j
// Values are the *unsigned* field values; the -2^(bits-1) of signed formats is applied by the
// caller as  sum(a*u) - 2^(bits-1) * sum(a)  (exact in int32) or with sub_bytes().
// ---------------------------------------------------------------------------------------------
template <int BITS>
struct I8Unpack {
  static constexpr int EPW = 32 / BITS;
  static constexpr int NQUAD = EPW / 4;  // == 8 / BITS
  static constexpr int field_of_slot(int x) { return (x & 3) * (8 / BITS) + (x >> 2); }
  static __device__ __forceinline__ void run(uint32_t w, uint32_t (&out)[NQUAD]) {
    constexpr uint32_t m = ((1u << BITS) - 1u) * 0x01010101u;
#pragma unroll
    for (int j = 0; j < NQUAD; ++j) out[j] = (w >> (BITS * j)) & m;
  }
};

// per-byte (u - z) for 0 <= u < 128, 0 <= z <= 127, no inter-byte borrow (SWAR)
__device__ __forceinline__ uint32_t sub_bytes(uint32_t u4, uint32_t z4) {
  return ((u4 | 0x80808080u) - z4) ^ 0x80808080u;
}

// per-byte (u - z) for codes u <= 3 (1- and 2-bit fields): ONE byte permute - the quad's bytes select from the table
// {0 - z, 1 - z, 2 - z, 3 - z} (built once from z with sub_bytes) - instead of the three operations of sub_bytes
__device__ __forceinline__ uint32_t sub_bytes_tbl(uint32_t u4, uint32_t tbl) {
  return __builtin_amdgcn_perm(0u, tbl, u4);
}

// per-byte (u - z) mod 256 (SWAR, no inter-byte borrow).  8-bit weights with quantized zero points: the
// TE expression subtracts in the int8 storage type (quantization.py:208-217), so the difference wraps
__device__ __forceinline__ uint32_t sub_bytes_mod(uint32_t u, uint32_t z) {
  constexpr uint32_t H = 0x80808080u;
  return ((u | H) - (z & ~H)) ^ ((u ^ ~z) & H);
}

}  // namespace wqaa
