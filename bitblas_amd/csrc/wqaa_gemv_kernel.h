// wqaa_gemv_kernel.h - W_q x A GEMV family for gfx950 (M < 8: the HBM-bound decode case).
//
// Replaces the reference's `GemvDequantizeSIMTScheduler` / `GemvFineGrainSIMTScheduler` kernel
// templates (bitblas/ops/general_matmul/tilelang/dequantize/gemv_dequantize_simt.py:83-262,
// tilelang/dense/gemv_simt.py:81-185).  Same computation,
//     C[m, n] = cast_out( sum_k A[m, k] * dq(B[n, k]) ) (+ Bias[n]),
// different machine mapping:
//   * one wave64 streams R weight rows; every lane issues 16-byte non-temporal loads, so a wave
//     instruction fetches 1 KiB of contiguous packed weights (the reference's default issues
//     4-byte loads per thread);
//   * a step = D lane-chunks x R rows of loads, all issued before anything is consumed; every load
//     is unconditional (out-of-range lanes/rows are clamped to a valid address and meet a
//     zero-filled activation slot), so the compiler can count them: the .s has `vmcnt(N)` ladders,
//     no `vmcnt(0)` between loads;
//   * the activation rows are staged once per workgroup into LDS as [piece][lane] 16-byte slots,
//     already permuted into the order the unpack produces (wqaa_decode.h), so ds_read_b128 is
//     conflict free and the inner loop has no shuffles.  Their global loads are issued BEFORE the
//     first weight step and written to LDS after it, so they never queue behind HBM traffic;
//   * everything that selects code (weight kind, layout, zero mode, batch tile) is a template
//     parameter: no runtime branch sits between loads;
//   * unpack -> (zero, scale) in packed fp16 exactly as the TE definition does it
//     (tirscript/matmul_dequantize_impl.py:391-451) -> V_DOT2_F32_F16 / V_DOT4_I32_I8 into
//     fp32 / int32 accumulators -> DPP row reduction + readlane.
#pragma once
#include "wqaa_common.h"
#include "wqaa_decode.h"
#include "wqaa_kinds.h"

namespace wqaa {

// ------------------------------------------------------------------------------------------
// wave reductions: DPP inside a 16-lane row, readlane across the four rows
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  v += dpp_f<0x140>(v);  // row_mirror -> every lane holds its 16-lane row sum
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int wave_sum(int v) {
  v += dpp_i<0xB1>(v);
  v += dpp_i<0x4E>(v);
  v += dpp_i<0x141>(v);
  v += dpp_i<0x140>(v);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// ------------------------------------------------------------------------------------------
// policy
// ------------------------------------------------------------------------------------------
// The reference computes `s = Qp / x.abs().max(...).clamp(min=1e-5)` with a Python int on the left
// (integration/BitNet/utils_quant.py:166): torch evaluates that as `reciprocal(t) * 127` - TWO fp32 roundings, which
// differs from the single quotient 127 / t in the last bit for about half of all inputs (pinned by
// tests/golden/bitnet_golden.npz, produced by running the reference function).  The reciprocal is taken in fp64 and
// rounded once: identical to a correctly rounded fp32 division (53 >= 2 * 24 + 2 bits).
__device__ inline float act_quant_scale(float absmax) {
  const float r = (float)(1.0 / (double)fmaxf(absmax, 1e-5f));
  return __fmul_rn(r, 127.0f);
}

template <int KIND_, int LAYOUT_, int AT_, int MB_, int MODE_, int FLAGS_, int R_ = 2, int D_ = 2, bool AD_ = false, bool KS_ = false>
struct GemvPolicy {
  static constexpr int KIND = KIND_, LAYOUT = LAYOUT_, AT = AT_, MB = MB_, MODE = MODE_, FLAGS = FLAGS_;
  static constexpr int R = R_;       // weight rows per wave per step
  static constexpr int D = D_;       // lane chunks per step
  static constexpr bool STRICT = (FLAGS_ & FL_STRICT) != 0;
  static constexpr bool A8 = (FLAGS_ & FL_A8) != 0;
  static constexpr bool BF = (FLAGS_ & FL_BF16) != 0;   // 16-bit float type is bfloat16
  // AD ("activations direct") members, M = 1 and K within one step (K <= D * 64 * E): every lane
  // reads its activation slice straight from global memory (L2-resident) into registers, once, and
  // keeps it for all the rows its wave visits: no LDS tile, no staging pass, no barrier.  Measured
  // against the LDS-staged form: 4096x4096 4.25 -> 4.11 us, 1024x1024 3.29 -> 2.81 us.  For longer K
  // the slice would have to be re-read every step (64 B of A per 32 B of weights through the same
  // texture path: 8192x28672 29.7 -> 30.5 us), so those and all batch tiles > 1 keep the LDS tile.
  static constexpr bool A4 = AT_ == AT_I4;              // packed int4 activations, widened while staged
  // BitNet layers (integration/BitNet/utils_quant.py:161-168, 205-216): the activations arrive as fp16 and
  // the workgroup applies the per-token absmax int8 quantiser itself while it stages them - the
  // caller's quantise -> matmul -> rescale chain is one launch
  static constexpr bool AQ = (FLAGS_ & FL_AQ) != 0 && AT_ == AT_I8;
  static constexpr bool AD = AD_ && MB_ <= 2 && !A8 && !A4 && !AQ;
  // KS: the member can split K across the waves of a workgroup (few-row shards).  A compile-time twin rather than a
  // run-time switch: the switch alone cost the unsplit LDS-staged member 4-7 % (11008x4096 7.4 -> 7.7 us, 28672x8192
  // 28.9 -> 30.9: different register allocation and scheduling of the hot loop), which the many-row shapes must not pay.
  static constexpr bool KS = KS_ && !AD && !AQ;
  using T = KindTraits<KIND_, AT_>;
  // words of raw activation data per staging item (one decode unit = G elements)
  static constexpr int AW = AT_ == AT_I4 ? T::G / 8 : AQ ? T::G / 2 : (AT_ == AT_I8 || A8) ? T::G / 4 : T::G / 2;
  // activation items per thread loaded ahead of the weights (<= 32 VGPRs)
  static constexpr int NA = 32 / AW > 8 ? 8 : (32 / AW < 1 ? 1 : 32 / AW);
};

template <class P>
struct Stage {
  u32x4 w[P::R];
  uint32_t s[P::R];  // scale bits (low 16)
  uint32_t z[P::R];  // zero bits (low 16) / raw qzeros byte
  uint32_t araw[P::AD ? P::MB : 1][P::AD ? P::T::UNITS * P::AW : 1];   // AD: the lane's activation slices (one per batch row), natural order
  bool avalid;
};

// AD members: activation piece `pp` of unit `u` in the order the unpack produces, built from the raw
// natural-order words with compile-time v_perm selectors (nothing for layouts whose order is natural)
template <class P, int PP, int Eo>
__device__ __forceinline__ uint32_t a_piece_word(const uint32_t* raw /* AW words of the unit */) {
  using T = typename P::T;
  if constexpr (P::AT == AT_F16) {
    constexpr int sa = T::src_elem(P::LAYOUT, PP * T::PE + 2 * Eo), sb = T::src_elem(P::LAYOUT, PP * T::PE + 2 * Eo + 1);
    if constexpr (sb == sa + 1 && (sa % 2) == 0) {
      return raw[sa / 2];
    } else {
      constexpr uint32_t sel = ((uint32_t)(4 + 2 * (sb % 2) + 1) << 24) | ((uint32_t)(4 + 2 * (sb % 2)) << 16) |
                               ((uint32_t)(2 * (sa % 2) + 1) << 8) | (uint32_t)(2 * (sa % 2));
      return __builtin_amdgcn_perm(raw[sb / 2], raw[sa / 2], sel);
    }
  } else {
    constexpr int s0 = T::src_elem(P::LAYOUT, PP * T::PE + 4 * Eo), s1 = T::src_elem(P::LAYOUT, PP * T::PE + 4 * Eo + 1),
                  s2 = T::src_elem(P::LAYOUT, PP * T::PE + 4 * Eo + 2), s3 = T::src_elem(P::LAYOUT, PP * T::PE + 4 * Eo + 3);
    if constexpr (s1 == s0 + 1 && s2 == s0 + 2 && s3 == s0 + 3 && (s0 % 4) == 0) {
      return raw[s0 / 4];
    } else {
      constexpr uint32_t selA = 0x0C0C0000u | ((uint32_t)(4 + (s1 % 4)) << 8) | (uint32_t)(s0 % 4);
      constexpr uint32_t selB = 0x0C0C0000u | ((uint32_t)(4 + (s3 % 4)) << 8) | (uint32_t)(s2 % 4);
      const uint32_t lo = __builtin_amdgcn_perm(raw[s1 / 4], raw[s0 / 4], selA);
      const uint32_t hi = __builtin_amdgcn_perm(raw[s3 / 4], raw[s2 / 4], selB);
      return lo | (hi << 16);
    }
  }
}
template <class P, int PP>
__device__ __forceinline__ u32x4 a_piece(const uint32_t* raw, bool valid) {
  u32x4 v = {a_piece_word<P, PP, 0>(raw), a_piece_word<P, PP, 1>(raw), a_piece_word<P, PP, 2>(raw), a_piece_word<P, PP, 3>(raw)};
  if (!valid) v = u32x4{0u, 0u, 0u, 0u};
  return v;
}
template <class P>
__device__ __forceinline__ u32x4 a_piece_rt(const uint32_t* raw, int pp, bool valid) {
  // pp is a fully unrolled loop index in the callers; the switch folds away
  switch (pp) {
    case 0: return a_piece<P, 0>(raw, valid);
    case 1: if constexpr (P::T::PU > 1) return a_piece<P, 1>(raw, valid); break;
    case 2: if constexpr (P::T::PU > 2) return a_piece<P, 2>(raw, valid); break;
    case 3: if constexpr (P::T::PU > 3) return a_piece<P, 3>(raw, valid); break;
  }
  return u32x4{0u, 0u, 0u, 0u};
}


__device__ __forceinline__ half_t fp8_to_half(uint8_t v, bool e5m2) {
  if (e5m2) return __builtin_bit_cast(half_t, (uint16_t)((uint16_t)v << 8));
  const uint16_t mag = (uint16_t)((v & 0x7Fu) << 7);
  half_t h = __builtin_bit_cast(half_t, mag) * (half_t)256.0f;
  const uint16_t bits = (uint16_t)(__builtin_bit_cast(uint16_t, h) | ((uint16_t)(v & 0x80u) << 8));
  return __builtin_bit_cast(half_t, bits);
}

template <int NWORDS>
__device__ __forceinline__ void load_words(const void* p, uint32_t (&w)[NWORDS]) {
  if constexpr (NWORDS % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NWORDS / 4; ++q) {
      const u32x4 v = reinterpret_cast<const u32x4*>(p)[q];
      w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
    }
  } else if constexpr (NWORDS == 2) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    w[0] = v[0]; w[1] = v[1];
  } else {
    static_assert(NWORDS == 1, "unsupported activation item width");
    w[0] = *reinterpret_cast<const uint32_t*>(p);
  }
}

// ---- activation staging: item idx = ((mi * ncp + c) * UNITS + u) * 64 + lane -----------------
template <class P>
struct AItem {
  uint32_t w[P::AW];
  bool valid;
};

template <class P>
__device__ __forceinline__ void a_item_load(const GemvArgs& a, int m0, int idx, AItem<P>& it) {
  using T = typename P::T;
  const int mi = idx % P::MB;          // all divisors are compile-time powers of two
  const int i = idx / P::MB;
  const int l = i & 63;
  const int u = (i >> 6) % T::UNITS;
  const int c = (i >> 6) / T::UNITS;
  const int kb = (c * 64 + l) * T::E + u * T::G;
  it.valid = kb < a.K && (m0 + mi) < a.m;
  const long off = it.valid ? (long)(m0 + mi) * a.K + kb : 0;   // clamped: always a readable address
  constexpr int esz = ((P::AT == AT_I8 && !P::AQ) || P::A8) ? 1 : 2;
  if constexpr (P::A4) load_words<P::AW>(reinterpret_cast<const uint8_t*>(a.A) + off / 2, it.w);   // two per byte
  else load_words<P::AW>(reinterpret_cast<const uint8_t*>(a.A) + off * esz, it.w);
}

template <class P>
__device__ __forceinline__ void a_item_store(const GemvArgs& a, int ncp, int idx, const AItem<P>& it, u32x4* a_lds,
                                             float aq_s = 0.f) {
  using T = typename P::T;
  constexpr int G = T::G, PE = T::PE, PU = T::PU, UNITS = T::UNITS, PIECES = T::PIECES;
  const int mi = idx % P::MB;
  const int i = idx / P::MB;
  const int l = i & 63;
  const int u = (i >> 6) % UNITS;
  const int t = mi * ncp + (i >> 6) / UNITS;
  if constexpr (P::AT == AT_F16) {
    half_t src[G];
    if constexpr (P::A8) {
#pragma unroll
      for (int e = 0; e < G; ++e)
        src[e] = fp8_to_half((uint8_t)(it.w[e / 4] >> (8 * (e % 4))), a.a_fmt == WQAA_E5M2);
    } else {
#pragma unroll
      for (int e = 0; e < G / 2; ++e) {
        const half2_t h = as_h2(it.w[e]);
        src[2 * e] = h[0];
        src[2 * e + 1] = h[1];
      }
    }
#pragma unroll
    for (int pp = 0; pp < PU; ++pp) {
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2_t h = {src[T::src_elem(P::LAYOUT, pp * PE + 2 * e)],
                           src[T::src_elem(P::LAYOUT, pp * PE + 2 * e + 1)]};
        out[e] = it.valid ? as_u32(h) : 0u;
      }
      a_lds[((long)t * PIECES + u * PU + pp) * 64 + l] = out;
    }
  } else {
    uint8_t src[G];
    if constexpr (P::AQ) {
      // q = clamp(round(x * s), -128, 127), round half to even (torch.round), utils_quant.py:165-167
#pragma unroll
      for (int e = 0; e < G; ++e) {
        const half2_t h = as_h2(it.w[e / 2]);
        float q = rintf((float)h[e & 1] * aq_s);
        q = fminf(fmaxf(q, -128.f), 127.f);
        src[e] = (uint8_t)(int)q;
      }
    } else if constexpr (P::A4) {
#pragma unroll
      for (int q = 0; q < G / 8; ++q) {
        uint32_t lo4, hi4;
        widen_nibbles(it.w[q], lo4, hi4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          src[8 * q + e] = (uint8_t)(lo4 >> (8 * e));
          src[8 * q + 4 + e] = (uint8_t)(hi4 >> (8 * e));
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < G; ++e) src[e] = (uint8_t)(it.w[e / 4] >> (8 * (e % 4)));
    }
#pragma unroll
    for (int pp = 0; pp < PU; ++pp) {
      u32x4 out;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v |= (uint32_t)src[T::src_elem(P::LAYOUT, pp * PE + 4 * q + e)] << (8 * e);
        out[q] = it.valid ? v : 0u;
      }
      a_lds[((long)t * PIECES + u * PU + pp) * 64 + l] = out;
    }
  }
}

template <class P>
__device__ __forceinline__ void decode_unit_f16(const u32x4& w, int u, half_t zf, const DecodeCtx& cx,
                                                const Lut16& lut, half2_t (&q)[P::T::G / 2]) {
  using T = typename P::T;
  constexpr int WPU = T::WPU;
  if constexpr (P::KIND == DK_INT1) {
    // int1 is a sign-extended 1-bit field (quantization.py:220-230): value = -u = (1 - u) - 1, so
    // invert the word and reuse the generic "field - 2^(bits-1)" path
    F16Unpack<T::BITS>::run(w[u] ^ cx.flip, zf, cx.magic, q);
  } else if constexpr (P::KIND == DK_INT4 || P::KIND == DK_INT2) {
    F16Unpack<T::BITS>::run(w[u], zf, cx.magic, q);
  } else if constexpr (P::KIND == DK_LUT4) {
    lut16_word(lut, w[u], q);
  } else if constexpr (P::KIND == DK_INT8) {
#pragma unroll
    for (int j = 0; j < WPU; ++j) {
      uint32_t x = w[u * WPU + j] ^ cx.flip;
      half2_t off = splat(cx.off8 + zf);
      if constexpr (P::MODE == MD_ZQ) {
        // (w - zero) in the int8 storage type: wraps mod 256, read back as a signed byte
        x = sub_bytes_mod(w[u * WPU + j], (uint32_t)(int)(float)zf * 0x01010101u) ^ 0x80808080u;
        off = splat((half_t)1152.0f);
      }
      const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, x, 0x04010400u);  // {b0,0x64,b1,0x64}
      const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, x, 0x04030402u);  // {b2,0x64,b3,0x64}
      q[2 * j] = as_h2(lo) - off;
      q[2 * j + 1] = as_h2(hi) - off;
    }
  } else if constexpr (P::KIND == DK_E4M3) {
#pragma unroll
    for (int j = 0; j < WPU; ++j) {
      half2_t t[2];
      unpack_e4m3_f16<P::STRICT>(w[u * WPU + j], t);
      q[2 * j] = t[0];
      q[2 * j + 1] = t[1];
    }
  } else if constexpr (P::KIND == DK_E5M2) {
#pragma unroll
    for (int j = 0; j < WPU; ++j) {
      half2_t t[2];
      unpack_e5m2_f16(w[u * WPU + j], t);
      q[2 * j] = t[0];
      q[2 * j + 1] = t[1];
    }
  } else {  // native fp16 weights
#pragma unroll
    for (int j = 0; j < WPU; ++j) q[j] = as_h2(w[u * WPU + j]);
  }
}

// The fields come out as unsigned bytes; the zero point of the signed formats (2^(bits-1), one value for the whole
// matrix: the reference defines no Scale / Zeros next to integer activations) is NOT subtracted per byte - the caller removes
// z * sum(a) from the row's sum instead (same int32 result, a third of the vector operations: the per-byte subtract was 3
// of the 6 operations per 4 weights)
template <class P>
__device__ __forceinline__ void decode_unit_i8(const u32x4& w, int u, uint32_t flip,
                                               uint32_t (&q)[P::T::G / 4]) {
  using T = typename P::T;
  if constexpr (T::SUBBYTE) {
    constexpr int NQ = I8Unpack<T::BITS>::NQUAD;
#pragma unroll
    for (int j = 0; j < T::WPU; ++j) {
      uint32_t t[NQ];
      // int1 signed: value = -u = (1 - u) - 1 -> invert the word first (see decode_unit_f16)
      I8Unpack<T::BITS>::run(w[u * T::WPU + j] ^ flip, t);
#pragma unroll
      for (int i = 0; i < NQ; ++i) q[j * NQ + i] = t[i];
    }
  } else {
#pragma unroll
    for (int j = 0; j < T::WPU; ++j) q[j] = w[u * T::WPU + j];
  }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
template <class P>
__global__ void __launch_bounds__(1024) wq_gemv_kernel(const GemvGroupArgs grp) {
  using T = typename P::T;
  const GemvArgs a = grp.p[blockIdx.z];          // kernel-argument segment, indexed by a dispatch-time scalar
  constexpr int R = P::R, MB = P::MB, D = P::D, NA = P::NA, MODE = P::MODE;
  constexpr int E = T::E, G = T::G, PU = T::PU, UNITS = T::UNITS, PIECES = T::PIECES;
  constexpr bool F16 = P::AT == AT_F16;
  using acc_t = typename std::conditional<F16, float, int>::type;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* a_lds = reinterpret_cast<u32x4*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nc = a.nc;                 // lane chunks (of 64 lanes) per row
  const int ncp = a.ncp;               // rounded up to a multiple of D: LDS slots beyond K are zero
  const int cpr = a.cpr;               // valid 16-byte lane chunks per row
  const int n_rg = (a.N + R - 1) / R;
  const int nthreads = blockDim.x;      // 64 / 128 / 256: picked by the selector
  const int NW = nthreads >> 6;
  // K split (few-row shards): kw consecutive waves share a row group
  const int kw = P::KS ? a.kw : 1;                // folds away in the unsplit members
  int wslot = wave, kpart = 0, slots = NW;
  if (kw > 1) {
    wslot = (int)(((uint32_t)wave * a.kw_magic) >> 16);
    kpart = wave - wslot * kw;
    slots = (int)(((uint32_t)NW * a.kw_magic) >> 16);
  }
  // this workgroup's row-group blocks (XCD-aware, wqaa_kinds.h); a workgroup without one leaves before any load
  const RowBlocks rb = xcd_row_blocks((int)blockIdx.x, (int)gridDim.x, a.n_rgb);
  if (rb.first >= rb.end) return;
  const int wg = rb.first * slots + wslot;
  const int rg_step = rb.stride * slots;
  const int m0 = blockIdx.y * MB;
  const uint8_t* Bp = reinterpret_cast<const uint8_t*>(a.B);
  const uint16_t* Sp = reinterpret_cast<const uint16_t*>(a.scale);
  const uint16_t* Zp = reinterpret_cast<const uint16_t*>(a.zeros);
  const uint8_t* Qp = reinterpret_cast<const uint8_t*>(a.zeros);
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;   // quantized zeros field width
  constexpr int ZPB = 8 / ZB;

  // ---- activations (batch tiles > 1): tiles larger than NA items/thread go through a plain loop first ----
  constexpr bool AD = P::AD;
  const int total_items = AD ? 0 : MB * ncp * 64 * UNITS;
  constexpr bool AQ = P::AQ;
  float aq_mx = 0.f;                      // AQ: max |x| over this thread's items (all of row tid % MB)
  auto item_absmax = [&](const AItem<P>& it) {
    if (!it.valid) return;
#pragma unroll
    for (int e = 0; e < P::AW; ++e) {
      const half2_t h = as_h2(it.w[e]);
      aq_mx = fmaxf(aq_mx, fmaxf(fabsf((float)h[0]), fabsf((float)h[1])));
    }
  };
  for (int idx = NA * nthreads + tid; idx < total_items; idx += nthreads) {
    AItem<P> it;
    a_item_load<P>(a, m0, idx, it);
    if constexpr (AQ) item_absmax(it);     // first pass: the scale needs the whole row; stored in the second pass
    else a_item_store<P>(a, ncp, idx, it, a_lds);
  }
  // the first NA items per thread: loads now (ahead of the weight stream), LDS writes after the
  // first weight step has been issued
  AItem<P> ahead[NA];
  if constexpr (!AD) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = j * nthreads + tid;
      if (j * nthreads < total_items)   // wave-uniform: whole rounds beyond the tile are skipped
        a_item_load<P>(a, m0, idx < total_items ? idx : 0, ahead[j]);
    }
  }
  const uint8_t* Arow = reinterpret_cast<const uint8_t*>(a.A) + (long)m0 * a.K * (F16 ? 2 : 1);

  // one step: D lane chunks x R rows, every load unconditional
  auto issue = [&](Stage<P> (&st)[D], int rg, int c0, bool load_a) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      int chunk = (c0 + d) * 64 + lane;
      if constexpr (AD) st[d].avalid = chunk < cpr;
      chunk = chunk < cpr ? chunk : 0;       // clamped lanes meet zero activations
      if (AD && load_a) {
        // the lane's E activations: UNITS loads of AW words, issued ahead of this chunk's weights
        constexpr int UB = T::G * (F16 ? 2 : 1);   // bytes per unit
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          // batch rows beyond m (a 2-row tile of a 1-row call never happens: MB follows m) are real rows
          const uint8_t* arow_mi = Arow + (long)mi * a.K * (F16 ? 2 : 1);
#pragma unroll
          for (int u = 0; u < UNITS; ++u) {
            uint32_t t[P::AW];
            load_words<P::AW>(arow_mi + (long)chunk * (UNITS * UB) + u * UB, t);
#pragma unroll
            for (int q = 0; q < P::AW; ++q) st[d].araw[mi][u * P::AW + q] = t[q];
          }
        }
      }
      // group of this lane chunk: chunk / (g / E), as a shift or a 32x32->hi multiply by ceil(2^32 / d)
      // (exact for chunk, d < 2^16), selected without a branch
      int gi = 0;
      if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (chunk >> a.gq_shift) : (int)__umulhi((uint32_t)chunk, a.gq_magic);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int n = rg * R + r;
        n = n < a.N ? n : a.N - 1;
        st[d].w[r] = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4*>(Bp + (long)n * a.row_bytes + (long)chunk * 16));
        if constexpr (MODE != MD_NONE) st[d].s[r] = Sp[(long)n * a.kg + gi];
        if constexpr (MODE == MD_ZO || MODE == MD_ZR) st[d].z[r] = Zp[(long)n * a.kg + gi];
        if constexpr (MODE == MD_ZQ) st[d].z[r] = Qp[(long)gi * a.zq_row_bytes + n / ZPB];
      }
    }
  };

  // this wave's lane-chunk range [c_lo, c_hi): all of the row, or its part of the K split
  const int c_lo = kw > 1 ? kpart * a.spp * D : 0;
  const int c_hi = kw > 1 ? (c_lo + a.spp * D < nc ? c_lo + a.spp * D : nc) : nc;
  Stage<P> st[D];
  int rg = wg;
  const bool have_work = kw > 1 || rg < n_rg;      // split workgroups iterate uniformly (barrier in finish), rows clamped
  issue(st, rg < n_rg ? rg : n_rg - 1, c_lo, true);

  float aq_s[MB];                         // AQ: act_quant_scale(max |row|) of every row of the batch tile
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) aq_s[mi] = 1.f;
  if constexpr (AQ) {
    // per-wave row maxima live behind the activation tile in the dynamic LDS block (the host adds 256 B)
    float* aq_wmax = reinterpret_cast<float*>(a_lds + (long)MB * ncp * T::PIECES * 64);
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (j * nthreads + tid < total_items) item_absmax(ahead[j]);
    // a thread's items all sit in row tid % MB (item strides are multiples of MB): butterfly over the
    // lanes of the same row, then across the waves through LDS
#pragma unroll
    for (int off = 32; off >= MB; off >>= 1) aq_mx = fmaxf(aq_mx, __shfl_xor(aq_mx, off));
    if (lane < MB) aq_wmax[wave * 4 + lane] = aq_mx;
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      float mx = 0.f;
      for (int w = 0; w < NW; ++w) mx = fmaxf(mx, aq_wmax[w * 4 + mi]);
      aq_s[mi] = act_quant_scale(mx);
    }
    float my_s = aq_s[0];
#pragma unroll
    for (int mi = 1; mi < MB; ++mi) my_s = (tid % MB) == mi ? aq_s[mi] : my_s;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = j * nthreads + tid;
      if (idx < total_items) a_item_store<P>(a, ncp, idx, ahead[j], a_lds, my_s);
    }
    for (int idx = NA * nthreads + tid; idx < total_items; idx += nthreads) {   // second pass over the tail (L2 hits)
      AItem<P> it;
      a_item_load<P>(a, m0, idx, it);
      a_item_store<P>(a, ncp, idx, it, a_lds, my_s);
    }
  } else if constexpr (!AD) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = j * nthreads + tid;
      if (idx < total_items) a_item_store<P>(a, ncp, idx, ahead[j], a_lds);
    }
  }

  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) {
      lut = make_fp4_lut(P::BF);
    } else {
      lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
    }
  }
  if constexpr (!AD) __syncthreads();

  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
  cx.flip = 0u;
  if (P::KIND == DK_INT1 && a.is_signed) cx.flip = 0xFFFFFFFFu;
  if (P::KIND == DK_INT8 && a.is_signed) cx.flip = 0x80808080u;
  // int4 x int4: the weight nibbles are two's complement, not offset binary: n ^ 8 is the offset code
  if (P::A4 && P::KIND == DK_INT4 && a.is_signed) cx.flip = 0x88888888u;
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  const int zpi = (!F16 && a.is_signed && T::SUBBYTE) ? (1 << (T::BITS - 1)) : 0;   // integer zero point of the sub-byte signed formats

  acc_t acc[R][MB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) acc[r][mi] = 0;
  int sa[MB];                                      // integer members: sum of the activations this lane has multiplied (per batch row)
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) sa[mi] = 0;
  int acc4[R][MB], acc16[R][MB];                   // 2-bit x int8: the sums of the fields taken at scale 4 and 16 (see consume)
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) acc4[r][mi] = acc16[r][mi] = 0;

  const bool need_mask = ncp * 64 != cpr;    // some lanes of the last step lie beyond K
  auto consume = [&](const Stage<P>& s, int c, int rg_now) {
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      if constexpr (F16) {
        half2_t q[R][G / 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          half_t zf = cx.zf;
          if constexpr (MODE == MD_ZQ) {
            // (w_u - zero_u) in the integer domain (quantization.py:208-217): ignores signedness
            const int n = rg_now * R + r;
            const uint32_t zq = (s.z[r] >> ((n % ZPB) * ZB)) & ((1u << ZB) - 1u);
            zf = (half_t)(float)zq;
          }
          if constexpr (P::BF) {
            // bf16: integer fields (minus the integer zero point) times the scale, one rounding
            if constexpr (P::KIND == DK_LUT4) {
              // nf4 / fp4: 16-bit table entries (bfloat16 bit patterns), then one rounding for the scale
              half2_t t[4];
              lut16_word(lut, s.w[r][u], t);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                q[r][i] = MODE == MD_NONE ? t[i] : as_h2(bf16x2_scale(as_u32(t[i]), bf16_bits_to_float(s.s[r])));
            } else if constexpr (T::SUBBYTE) {
              uint32_t pk[G / 2];
              constexpr int ZM = MODE == MD_ZO ? 1 : MODE == MD_ZR ? 2 : 0;
              unpack_word_bf16<T::BITS, 0, ZM>(s.w[r][u] ^ (P::KIND == DK_INT1 ? cx.flip : 0u), (float)zf,
                                               bf16_bits_to_float(s.s[r]), MODE != MD_NONE, pk,
                                               ZM ? bf16_bits_to_float(s.z[r]) : 0.f);
#pragma unroll
              for (int i = 0; i < G / 2; ++i) q[r][i] = as_h2(pk[i]);
            } else {
#pragma unroll
              for (int j = 0; j < T::WPU; ++j) {
                const uint32_t x = s.w[r][u * T::WPU + j];
                if constexpr (P::KIND == DK_INT8) {
                  const float sc = bf16_bits_to_float(s.s[r]);
                  float v[4];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const int b8 = (int)((x >> (8 * e)) & 0xFFu);
                    if constexpr (MODE == MD_ZQ) v[e] = (float)(int)(int8_t)(b8 - (int)(float)zf);   // int8 storage arithmetic wraps
                    else v[e] = (float)(a.is_signed ? (int)(int8_t)b8 : b8);
                  }
                  {
                    constexpr int ZM = MODE == MD_ZO ? 1 : MODE == MD_ZR ? 2 : 0;
                    const float zv = ZM ? bf16_bits_to_float(s.z[r]) : 0.f;
                    dequant_pair_bf16<ZM>(v[0], v[1], 0.f, sc, zv, MODE != MD_NONE);
                    dequant_pair_bf16<ZM>(v[2], v[3], 0.f, sc, zv, MODE != MD_NONE);
                  }
                  q[r][2 * j] = as_h2(cvt_pk_bf16(v[0], v[1]));
                  q[r][2 * j + 1] = as_h2(cvt_pk_bf16(v[2], v[3]));
                } else if constexpr (P::KIND == DK_E4M3) {
                  // e4m3 -> fp16 is exact (IEEE decode: the reference has no bfloat16 variant of its bit
                  // trick, quantization.py:169-176 asserts float16), fp16 -> bf16 of an e4m3 value too
                  half2_t t[2];
                  unpack_e4m3_f16<false>(x, t);
                  const float sc = MODE != MD_NONE ? bf16_bits_to_float(s.s[r]) : 1.f;
                  q[r][2 * j] = as_h2(cvt_pk_bf16((float)t[0][0] * sc, (float)t[0][1] * sc));
                  q[r][2 * j + 1] = as_h2(cvt_pk_bf16((float)t[1][0] * sc, (float)t[1][1] * sc));
                } else {
                  q[r][j] = as_h2(x);   // native bf16 weights
                }
              }
            }
          } else {
          decode_unit_f16<P>(s.w[r], u, zf, cx, lut, q[r]);
          if constexpr (MODE == MD_S || MODE == MD_ZQ) {
            const half2_t s2 = splat(bits_to_half(s.s[r]));
#pragma unroll
            for (int i = 0; i < G / 2; ++i) q[r][i] = q[r][i] * s2;
          } else if constexpr (MODE == MD_ZO) {
            const half2_t z2 = splat(bits_to_half(s.z[r])), s2 = splat(bits_to_half(s.s[r]));
#pragma unroll
            for (int i = 0; i < G / 2; ++i) q[r][i] = (q[r][i] - z2) * s2;
          } else if constexpr (MODE == MD_ZR) {
            const half2_t z2 = splat(bits_to_half(s.z[r])), s2 = splat(bits_to_half(s.s[r]));
#pragma unroll
            for (int i = 0; i < G / 2; ++i) {
              half2_t t = q[r][i] * s2;
              // keep the two roundings of `w * Scale - Zeros` (no contraction into an fma)
              asm volatile("" : "+v"(t));
              q[r][i] = t - z2;
            }
          }
          }  // !BF
        }
#pragma unroll
        for (int pp = 0; pp < PU; ++pp) {
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            u32x4 av;
            if constexpr (AD) {
              av = a_piece_rt<P>(s.araw[mi] + u * P::AW, pp, true);
              if (need_mask && !s.avalid) av = u32x4{0u, 0u, 0u, 0u};   // wave-uniform outer test: free when K has no ragged chunk
            } else {
              av = a_lds[((long)(mi * ncp + c) * PIECES + u * PU + pp) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if constexpr (P::BF)
                  acc[r][mi] = __builtin_amdgcn_fdot2_f32_bf16(as_bf2(as_u32(q[r][pp * 4 + e])), as_bf2(av[e]), acc[r][mi], false);
                else
                  acc[r][mi] = __builtin_amdgcn_fdot2(q[r][pp * 4 + e], as_h2(av[e]), acc[r][mi], false);
              }
          }
        }
      } else {
        uint32_t q[R][G / 4];
        // 2-bit fields: the four fields of a byte are taken WHERE THEY SIT (AND only; one shift brings the top field below the sign
        // bit) - they come out scaled by 1, 4, 16, 16 and go to one accumulator per scale, recombined with exact shifts when the
        // row is finished: 5 vector operations per 16 weights instead of 7
        constexpr bool CLS2 = T::BITS == 2 && P::AT == AT_I8 && G == 16;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if constexpr (CLS2) {
            const uint32_t w = s.w[r][u] ^ cx.flip;
            q[r][0] = w & 0x03030303u;
            q[r][1] = w & 0x0C0C0C0Cu;
            q[r][2] = w & 0x30303030u;
            q[r][3] = (w >> 2) & 0x30303030u;
          } else {
            decode_unit_i8<P>(s.w[r], u, cx.flip, q[r]);
          }
        }
#pragma unroll
        for (int pp = 0; pp < PU; ++pp) {
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            u32x4 av;
            if constexpr (AD) {
              av = a_piece_rt<P>(s.araw[mi] + u * P::AW, pp, true);
              if (need_mask && !s.avalid) av = u32x4{0u, 0u, 0u, 0u};   // wave-uniform outer test: free when K has no ragged chunk
            } else {
              av = a_lds[((long)(mi * ncp + c) * PIECES + u * PU + pp) * 64 + lane];
            }
            if constexpr (T::SUBBYTE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) sa[mi] = __builtin_amdgcn_sdot4((int)av[e], 0x01010101, sa[mi], false);
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if constexpr (CLS2) {
                  if (e == 0) acc[r][mi] = __builtin_amdgcn_sdot4((int)av[e], (int)q[r][e], acc[r][mi], false);
                  else if (e == 1) acc4[r][mi] = __builtin_amdgcn_sdot4((int)av[e], (int)q[r][e], acc4[r][mi], false);
                  else acc16[r][mi] = __builtin_amdgcn_sdot4((int)av[e], (int)q[r][e], acc16[r][mi], false);
                } else {
                  acc[r][mi] = __builtin_amdgcn_sdot4((int)av[e], (int)q[r][pp * 4 + e], acc[r][mi], false);
                }
              }
          }
        }
      }
    }
  };

  int it_idx = 0;
  auto finish = [&](int rg_now) {
    if constexpr (!F16 && T::SUBBYTE) {
      // sum_k (q - z) a = sum_k q a - z sum_k a over the chunks this lane multiplied (exact in int32)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if constexpr (T::BITS == 2 && P::AT == AT_I8 && G == 16) {
            acc[r][mi] += (acc4[r][mi] >> 2) + (acc16[r][mi] >> 4);     // multiples of 4 / 16: exact
            acc4[r][mi] = acc16[r][mi] = 0;
          }
          acc[r][mi] -= zpi * sa[mi];
        }
        sa[mi] = 0;
      }
    }
    if constexpr (P::KS) {
      if (kw > 1) {
        // the kw parts of a row group meet in LDS (double buffered: one barrier per row group) and are summed by part
        // 0 in part order - bit-identical from run to run, bit-exact for the integer members
        acc_t* red = reinterpret_cast<acc_t*>(a_lds + (long)MB * ncp * T::PIECES * 64) + (it_idx & 1) * (NW * R * MB);
        ++it_idx;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            const acc_t tot = wave_sum(acc[r][mi]);
            acc[r][mi] = 0;
            if (lane == 0) red[(wave * R + r) * MB + mi] = tot;
          }
        __syncthreads();
        if (kpart == 0 && lane == 0 && rg_now < n_rg) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int n = rg_now * R + r;
            if (n >= a.N) continue;
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
              if ((m0 + mi) >= a.m) continue;
              acc_t tot = red[(wave * R + r) * MB + mi];
              for (int p = 1; p < kw; ++p) tot += red[((wave + p) * R + r) * MB + mi];
              if constexpr (F16) {
                float b = 0.f;
                if (a.has_bias) b = P::BF ? bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[n])
                                         : (float)reinterpret_cast<const half_t*>(a.bias)[n];
                store_out(a.C, (long)(m0 + mi) * a.N + n, tot, a.out_dtype, a.has_bias != 0, b);
              } else if (a.epi_row) {
                store_out_fused(a.C, (long)(m0 + mi) * a.N + n, tot, a.epi_row[m0 + mi], a.epi_tensor, a.has_bias != 0, a.bias, n);
              } else {
                const int b = a.has_bias ? (int)reinterpret_cast<const int8_t*>(a.bias)[n] : 0;
                store_out(a.C, (long)(m0 + mi) * a.N + n, tot, a.out_dtype, a.has_bias != 0, b);
              }
            }
          }
        }
        return;
      }
    }
    if constexpr (F16 && R == 2 && !P::BF) {
      // the two rows of the group are neighbours in C: one 4- / 8-byte store per batch row instead of two stores
      // (measured on the exact-product members, wqaa_gemvx_kernel.h: 0.15-0.2 us per launch)
      const int n = rg_now * 2;
      if (n + 1 < a.N && (a.out_dtype == WQAA_F16 || a.out_dtype == WQAA_F32) && ((a.N & 1) == 0)) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          const float t0 = wave_sum(acc[0][mi]), t1 = wave_sum(acc[1][mi]);
          acc[0][mi] = 0;
          acc[1][mi] = 0;
          if (lane == 0 && (m0 + mi) < a.m) {
            const long idx = (long)(m0 + mi) * a.N + n;
            float b0 = 0.f, b1 = 0.f;
            if (a.has_bias) {
              b0 = (float)reinterpret_cast<const half_t*>(a.bias)[n];
              b1 = (float)reinterpret_cast<const half_t*>(a.bias)[n + 1];
            }
            if (a.out_dtype == WQAA_F16) {
              half_t h0 = (half_t)t0, h1 = (half_t)t1;
              if (a.has_bias) { h0 = h0 + (half_t)b0; h1 = h1 + (half_t)b1; }
              *reinterpret_cast<uint32_t*>(reinterpret_cast<half_t*>(a.C) + idx) = as_u32(half2_t{h0, h1});
            } else {
              float2_t v = {t0, t1};
              if (a.has_bias) { v[0] += b0; v[1] += b1; }
              *reinterpret_cast<float2_t*>(reinterpret_cast<float*>(a.C) + idx) = v;
            }
          }
        }
        return;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int n = rg_now * R + r;
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) {
        const acc_t tot = wave_sum(acc[r][mi]);
        acc[r][mi] = 0;
        if (lane == 0 && n < a.N && (m0 + mi) < a.m) {
          if constexpr (F16) {
            float b = 0.f;
            if (a.has_bias) b = P::BF ? bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[n])
                                     : (float)reinterpret_cast<const half_t*>(a.bias)[n];
            store_out(a.C, (long)(m0 + mi) * a.N + n, tot, a.out_dtype, a.has_bias != 0, b);
          } else {
            if constexpr (AQ) {
              store_out_fused(a.C, (long)(m0 + mi) * a.N + n, tot, aq_s[mi], a.epi_tensor, a.has_bias != 0, a.bias, n);
            } else if (a.epi_row) {
              store_out_fused(a.C, (long)(m0 + mi) * a.N + n, tot, a.epi_row[m0 + mi], a.epi_tensor, a.has_bias != 0, a.bias, n);
            } else {
              const int b = a.has_bias ? (int)reinterpret_cast<const int8_t*>(a.bias)[n] : 0;
              store_out(a.C, (long)(m0 + mi) * a.N + n, tot, a.out_dtype, a.has_bias != 0, b);
            }
          }
        }
      }
    }
  };

  if (!have_work) return;
  // first step was issued before the barrier; later steps are issued right after the previous
  // step's registers are consumed
  int c0 = c_lo;
  int it_left = 1;                                 // blocks of this workgroup (uniform: the K-split parts meet behind a barrier)
  if (rb.first + rb.stride < rb.end) it_left = (rb.end - rb.first + rb.stride - 1) / rb.stride;
  while (true) {
#pragma unroll
    for (int d = 0; d < D; ++d) consume(st[d], c0 + d, rg);
    c0 += D;
    if (c0 >= c_hi) {
      finish(rg);
      c0 = c_lo;
      rg += rg_step;
      if (--it_left <= 0 || (kw == 1 && rg >= n_rg)) break;
    }
    issue(st, rg, c0, false);   // AD members: K fits one step, the activation registers stay as loaded
  }
}

// ------------------------------------------------------------------------------------------
// member tables.  The family is instantiated in wqaa_gemv_inst_*.hip, one translation unit per group of members
// (parallel builds; a probe can include this header and instantiate a single member in seconds).
// ------------------------------------------------------------------------------------------
typedef void (*gemv_fn)(const GemvGroupArgs);

static constexpr int kDirectTile = 101;   // pick_mb code of the M = 1 "activations direct" member
static constexpr int kSplitTile = 200;    // + mb (1, 2): the K-split twins of the LDS-staged members
// + mb (1, 2, 4): four rows per wave, ONE lane chunk per step - for rows with an odd number of lane chunks (2-bit weights at
// K = 4096 are exactly one: 64 lanes x 64 weights).  The (2, 2) members pad such rows to an even count: a second load of
// chunk 0 and a decode against zeros, half the work of a one-chunk row.  Integer activations only (BASELINE c4, BitNet).
static constexpr int kChunkTile = 300;
static constexpr int kChunkDirect = 310;  // + mb (1, 2): ... with the activations in registers
static constexpr int kBatchTiles[] = {1, kDirectTile, kDirectTile + 1, kDirectTile + 2, 2, 4, kSplitTile + 1, kSplitTile + 2,
                                      kChunkTile + 1, kChunkTile + 2, kChunkTile + 4, kChunkDirect + 1, kChunkDirect + 2};

// The register-resident ("direct") members exist only where the activation slice of a lane chunk fits the register file:
// mb * E elements of 2 (1) bytes <= 128 bytes for the 64 dwords a 128-VGPR kernel can spare.  4-bit weights pass at M <= 2,
// 2-bit x fp16 at M = 1, 1-bit x fp16 never: those compiled to 440 - 3400 B of scratch per lane with the reloads inside the row
// loop and ran 5 - 23 x slower than their LDS-staged twins (profiles/r03_ab_direct_fit.txt: uint1 x fp16 M = 2 4096^2 220.6 vs
// 9.5 us, uint2 51.0 vs 6.0, int1 x int8 40.8 vs 6.8) - they are not built, and gemv_choose asks for the same condition.
template <int KIND, int AT, int MB, int FLAGS = 0>
constexpr bool gemv_direct_fits() {
  // (the table-decoded 4-bit formats with bfloat16 activations at M = 2 need the 16 table registers and the widening on
  // top of a full slice: 348 B of scratch, profiles/r03_static_isa.txt - LDS-staged as well)
  if (KIND == DK_LUT4 && MB == 2 && (FLAGS & FL_BF16) != 0) return false;
  return MB * (128 / KindTraits<KIND, AT>::BITS) * (AT == AT_F16 ? 2 : 1) <= 128;
}

// ... and where GemvPolicy::AD holds at all: fp8 / packed int4 activations and the in-kernel quantiser stage through LDS (round 6: a
// census of the built library found 31 such instantiations - twins of the LDS-staged members under another name, asked for by no rule)
template <int KIND, int AT, int MB, int FLAGS = 0>
constexpr bool gemv_direct_ok() {
  return gemv_direct_fits<KIND, AT, MB, FLAGS>() && (FLAGS & (FL_A8 | FL_AQ)) == 0 && AT != AT_I4;
}

template <int KIND, int LAYOUT, int AT, int MODE, int FLAGS>
static gemv_fn pick_mb(int mb) {
  switch (mb) {
    case 1: return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS>>;
    case kDirectTile:
      if constexpr (gemv_direct_ok<KIND, AT, 1, FLAGS>()) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS, 2, 2, true>>;
      else return nullptr;
    case kDirectTile + 1:
      if constexpr (gemv_direct_ok<KIND, AT, 1, FLAGS>()) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS, 1, 2, true>>;
      else return nullptr;
    case kDirectTile + 2:
      if constexpr (gemv_direct_ok<KIND, AT, 2, FLAGS>()) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 2, MODE, FLAGS, 2, 2, true>>;
      else return nullptr;
    case 2: return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 2, MODE, FLAGS>>;
    // (the 4-row batch tile runs where the MFMA family refuses a shape - N off its multiple of 4, K off its k-step: wqaa_abi.hip dispatch;
    // 1-bit weights with packed zero points never get there: a lane chunk is a whole k-step, and N x 1 bit fills whole bytes)
    case 4:
      if constexpr (KIND == DK_INT1 && MODE == MD_ZQ && AT == AT_F16) return nullptr;
      else return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 4, MODE, FLAGS>>;
    // (the in-kernel quantiser has no K-split twin: GemvPolicy::KS, and gemv_choose never asks)
    case kSplitTile + 1:
      if constexpr ((FLAGS & FL_AQ) == 0) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS, 2, 2, false, true>>;
      else return nullptr;
    case kSplitTile + 2:
      if constexpr ((FLAGS & FL_AQ) == 0) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 2, MODE, FLAGS, 2, 2, false, true>>;
      else return nullptr;
    case kChunkTile + 1:
      if constexpr (AT == AT_I8 && KindTraits<KIND, AT>::SUBBYTE) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS, 4, 1>>;
      else return nullptr;
    case kChunkTile + 2:
      if constexpr (AT == AT_I8 && KindTraits<KIND, AT>::SUBBYTE) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 2, MODE, FLAGS, 4, 1>>;
      else return nullptr;
    case kChunkTile + 4:
      if constexpr (AT == AT_I8 && KindTraits<KIND, AT>::SUBBYTE) return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 4, MODE, FLAGS, 4, 1>>;
      else return nullptr;
    case kChunkDirect + 1:
      if constexpr (AT == AT_I8 && KindTraits<KIND, AT>::SUBBYTE && (FLAGS & (FL_A8 | FL_AQ)) == 0 && gemv_direct_fits<KIND, AT, 1>())
        return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 1, MODE, FLAGS, 4, 1, true>>;
      else return nullptr;
    case kChunkDirect + 2:
      if constexpr (AT == AT_I8 && KindTraits<KIND, AT>::SUBBYTE && (FLAGS & (FL_A8 | FL_AQ)) == 0 && gemv_direct_fits<KIND, AT, 2>())
        return wq_gemv_kernel<GemvPolicy<KIND, LAYOUT, AT, 2, MODE, FLAGS, 4, 1, true>>;
      else return nullptr;
    default: return nullptr;
  }
}

template <int KIND, int LAYOUT>
static gemv_fn pick_mode_f16(int mode, int mb) {
  switch (mode) {
    case MD_NONE: return pick_mb<KIND, LAYOUT, AT_F16, MD_NONE, 0>(mb);
    case MD_S: return pick_mb<KIND, LAYOUT, AT_F16, MD_S, 0>(mb);
    case MD_ZO: return pick_mb<KIND, LAYOUT, AT_F16, MD_ZO, 0>(mb);
    case MD_ZR: return pick_mb<KIND, LAYOUT, AT_F16, MD_ZR, 0>(mb);
    case MD_ZQ: return pick_mb<KIND, LAYOUT, AT_F16, MD_ZQ, 0>(mb);
    default: return nullptr;
  }
}

// formats the reference never pairs with zero points: plain and scaled members only
template <int KIND, int FLAGS>
static gemv_fn pick_mode_fp(int mode, int mb) {
  switch (mode) {
    case MD_NONE: return pick_mb<KIND, LAYOUT_PLAIN, AT_F16, MD_NONE, FLAGS>(mb);
    case MD_S: return pick_mb<KIND, LAYOUT_PLAIN, AT_F16, MD_S, FLAGS>(mb);
    default: return nullptr;
  }
}


gemv_fn pick_gemv_f16_int(int kind, int layout, int mode, int mb);          // uint/int 4, 2, 1 x fp16, both layouts, all modes
gemv_fn pick_gemv_f16_other(int kind, int mode, int flags, int mb);         // int8, nf4/fp4, e4m3, e5m2, native fp16, dense fp8
gemv_fn pick_gemv_bf16(int kind, int mode, int mb);                         // bfloat16 activations (plain layout)
gemv_fn pick_gemv_int(int kind, int layout, int at, int flags, int mb);     // int8 / packed int4 activations (+ in-kernel quantiser)

}  // namespace wqaa
