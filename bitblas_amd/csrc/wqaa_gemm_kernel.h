// wqaa_gemm_kernel.h - W_q x A MFMA GEMM family for gfx950 (M >= 8: the matrix-core-bound case).
//
// Replaces the reference's tensor-core templates `MatmulDequantizeMMAScheduler` /
// `MatmulMMAScheduler` (bitblas/ops/general_matmul/tilelang/dequantize/matmul_dequantize_mma.py:200-508,
// tilelang/dense/matmul_mma.py:145-320).  Same computation as the GEMV family,
//     C[m, n] = cast_out( sum_k A[m, k] * dq(B[n, k]) ) (+ Bias[n]),
// but the machine mapping is built around the CDNA4 matrix core:
//   * packed weights never touch LDS.  The reference moves B global -> smem (packed) -> registers ->
//     dequantise -> smem (fp16) -> ldmatrix -> mma (two shared-memory round trips).  Here a lane owns
//     row n = lane & 15 of a 16-row fragment and the k-block kb = lane >> 4, exactly the operand map of
//     v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8, so ONE 16-byte load per lane (32 int4 weights)
//     is unpacked + (zero, scale)-dequantised in registers straight into the operands of FOUR MFMAs
//     of a 128-deep k-step.  The sum over k is order-free, so "k-block kb" is free to mean "the
//     lane's 32 consecutive k": no shuffles, no permuted checkpoint layout;
//   * the weight fragment is the MFMA *A* operand and the activation fragment the *B* operand
//     (D = W_frag x A_frag^T): a lane then owns 4 consecutive n of one output row m, i.e. one
//     8-byte fp16 store instead of four 2-byte ones;
//   * activations are staged global -> registers -> LDS (double buffered, one barrier per k-step) as
//     [row][16 granules of 16 B]; granule (kb, j) of row r lives in physical slot ((j*4+kb) ^ (r&15)):
//     both the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups hit 16 distinct slots;
//   * a workgroup is 4 waves side by side along N; every wave multiplies the whole BM x 128
//     activation tile by its own NFW weight fragments, so no weight word is decoded twice;
//   * the dequant arithmetic is the TE definition's (tirscript/matmul_dequantize_impl.py:391-451),
//     rounding in A_dtype per element; accumulation is fp32 / int32 in the matrix core.
#pragma once
#include "wqaa_common.h"
#include "wqaa_decode.h"
#include "wqaa_kinds.h"

#include <mutex>
#include <type_traits>
#include <utility>

#ifndef WQ_SETPRIO
#define WQ_SETPRIO 0   /* measured: -20 % at 256x256 (both waves of a SIMD raise priority together) */
#endif

namespace wqaa {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const void* A;
  const void* B;
  const void* lut;
  const void* scale;
  const void* zeros;
  const void* bias;
  void* C;
  int M, N, K;
  int kg;             // groups per weight row
  int gq_shift;       // 32-element k-blocks per group = g / KL as a shift (-1: use gq_magic)
  uint32_t gq_magic;
  long row_bytes;     // bytes per weight row
  int has_bias, out_dtype, is_signed, fp4_table;
  int zq_row_bytes;
  int tiles_m, tiles_n;
  int nsteps;         // K / KS
  int group_m;        // M-tiles per group of the tile order (1 = row-major)
  const float* epi_row;   // fused caller epilogue (wqaa_matmul_ex): out = half(acc / epi_row[m] / epi_tensor)
  float epi_tensor;
  int ksplit;         // > 1: workgroup (tile, s) covers k-steps [s*nsteps/ksplit, (s+1)*nsteps/ksplit) and
  void* ws;           //      writes fp32 / int32 partial sums to ws[s][M][N]; a second kernel reduces
  int ws_policy;      // bits 0-1: cache policy of the partial-sum stores (0 default, 1 non-temporal, 2 sc1, 3 sc0 sc1: write-through);
                      // bit 4: the ping-pong members store their output tile write-through (large outputs)
  // reciprocals of the tile map's divisors (tile_magic; 0 = divide): a uniform integer division is ~25 instructions of
  // float reciprocal + fix-up on this ISA (a 64-bit one ~150) and the tile map had five of them in front of the first load
  uint32_t mg_ntiles = 0, mg_per_group = 0, mg_group_m = 0, mg_tail_m = 0, mg_ksplit = 0;
  int decode_long = 0; // one-launch decode member: the wave's WHOLE k-range of the activations fits its LDS region in M-sized slots (wq_gemm_decode_lds_kernel)
  int tile_n_off = 0;  // ping-pong members: first N-tile of this launch (a launch may cover a band of the output's columns: the
                       // remainder of a partial round goes out as a second launch of the 128-row tile, csrc/wqaa_gemm.hip)
  // mid-M member (wqaa_gemm_mid_kernel.h): the tiles' sync words (library-owned, zero between launches) and the bound of the
  // in-launch wait in 10 ns ticks (0: nobody waits - every portion goes through the abandon / sweep path; test aid)
};

// floor(2^32 / d) + 1: __umulhi(x, magic) == x / d whenever x * d < 2^32 (the host checks the largest x it can meet)
inline uint32_t tile_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }
__host__ __device__ __forceinline__ int udiv_magic(int x, int d, uint32_t magic) {
  if (d == 1) return x;
#if defined(__HIP_DEVICE_COMPILE__)
  return magic ? (int)__umulhi((uint32_t)x, magic) : x / d;
#else
  return magic ? (int)(((unsigned long long)(uint32_t)x * magic) >> 32) : x / d;      // host twin (wqaa_debug_tile_of_block)
#endif
}

// tile of a workgroup.  XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous range of the grouped order
// (consecutive tile ids sweep `group_m` M-tiles x all N-tiles column by column, so the ~32 tiles an XCD has in flight form a
// compact 2-D block and share both operand bands in its L2); with split-K the k-slice is the slowest index.
struct TileOfBlock {
  int split, tile_m, tile_n;
};
__host__ __device__ __forceinline__ TileOfBlock tile_of_block(const GemmArgs& a, int block, int nblocks) {
  int blk = block;
  if ((nblocks & 7) == 0) blk = (block & 7) * (nblocks >> 3) + (block >> 3);
  TileOfBlock t;
  const int ntiles = a.tiles_m * a.tiles_n;
  t.split = a.ksplit > 1 ? udiv_magic(blk, ntiles, a.mg_ntiles) : 0;        // k-slice of this workgroup
  blk -= t.split * ntiles;
  const int per_group = a.group_m * a.tiles_n;
  const int grp = udiv_magic(blk, per_group, a.mg_per_group);
  const int rem = blk - grp * per_group;
  const int first_m = grp * a.group_m;
  const bool tail = a.tiles_m - first_m < a.group_m;                          // the last, shorter group of M-tiles
  const int gsz = tail ? a.tiles_m - first_m : a.group_m;
  t.tile_n = udiv_magic(rem, gsz, tail ? a.mg_tail_m : a.mg_group_m);
  t.tile_m = first_m + rem - t.tile_n * gsz;
  return t;
}

// lab builds only (tools/decode_trace.hip, -DWQAA_TRACE): per-wave timestamps kept in registers, written through a.lut
// (unused by the integer formats the harness instantiates) after the last phase
#ifdef WQAA_TRACE
#define WQ_TRACE_DECL unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long tr_real0_ = __builtin_amdgcn_s_memrealtime()
#define WQ_TRACE(i) tr_[i] = __builtin_readcyclecounter()
#define WQ_TRACE_IF(c, i) if (c) tr_[i] = __builtin_readcyclecounter()
#define WQ_TRACE_WAIT_BEGIN const unsigned long long tw0_ = __builtin_readcyclecounter()
#define WQ_TRACE_WAIT_END(i) tr_[i] += __builtin_readcyclecounter() - tw0_
#define WQ_TRACE_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WQ_TRACE_DUMP(nw)                                                                                             \
  if (lane == 0 && a.lut) {                                                                                             \
    unsigned long long* d_ = reinterpret_cast<unsigned long long*>(const_cast<void*>(a.lut)) + ((long)blockIdx.x * (nw) + wave) * 16;      \
    for (int i_ = 0; i_ < 8; ++i_) d_[i_] = tr_[i_];                                                                   \
    d_[8] = tr_real0_;                                                                                                 \
    d_[9] = __builtin_amdgcn_s_memrealtime();                                                                          \
  }
#else
#define WQ_TRACE_DECL
#define WQ_TRACE(i)
#define WQ_TRACE_IF(c, i)
#define WQ_TRACE_WAIT_BEGIN
#define WQ_TRACE_WAIT_END(i)
#define WQ_TRACE_DRAIN
#define WQ_TRACE_DUMP(nw)
#endif

// ------------------------------------------------------------------------------------------
// policy: one k-step is KS = 4 * KL deep; a lane owns KL consecutive k of one weight row
// ------------------------------------------------------------------------------------------
template <int KIND_, int LAYOUT_, int AT_, int MODE_, int FLAGS_, int MF_, int NWAVES_ = 4, int NFW_ = 2, int SK_ = 0, bool WIDE_ = false>
struct GemmPolicy {
  // WIDE: Scale / Zeros of four consecutive groups per 8-byte load (the host picks these members when a k-step is
  // exactly one group and K / g is a multiple of 4)
  static constexpr bool WIDE = WIDE_ && (MODE_ == MD_S || MODE_ == MD_ZO || MODE_ == MD_ZR) && SK_ == 0;
  // SK > 0: "skinny" member for decode batches - a workgroup owns SK consecutive k-steps of its tile,
  // issues ALL their weight loads before anything is consumed and stages all SK activation tiles behind
  // one barrier (the pipelined member is latency-bound when M is small: one HBM round trip per k-step)
  static constexpr int SK = SK_;
  static constexpr int KIND = KIND_, LAYOUT = LAYOUT_, AT = AT_, MODE = MODE_, FLAGS = FLAGS_;
  static constexpr int MF = MF_;        // 16-row activation fragments per workgroup (BM = 16 * MF)
  static constexpr int NFW = NFW_;      // 16-row weight fragments per wave
  static constexpr int NWAVES = NWAVES_; // waves side by side along N
  static constexpr int THREADS = 64 * NWAVES_;
  static constexpr int AG = (16 * MF_ * 16) / (64 * NWAVES_);   // activation granules per thread per k-step
  static constexpr bool DECODE = NWAVES_ == 8 && NFW_ == 1;   // decode-batch member: no LDS staging at all
  static_assert(DECODE || (AG >= 1 && AG * 64 * NWAVES_ == 16 * MF_ * 16), "tile / workgroup mismatch");
  static constexpr int BM = 16 * MF, BN = 16 * NFW * NWAVES;
  static constexpr bool STRICT = (FLAGS_ & FL_STRICT) != 0;
  static constexpr bool BF = (FLAGS_ & FL_BF16) != 0;   // 16-bit float type is bfloat16
  using T = KindTraits<KIND_, AT_>;
  static constexpr int BITS = T::BITS;
  static constexpr int EPW = T::EPW;
  // elements of k per lane per MFMA, MFMAs per k-step, k per lane per k-step
  static constexpr int KPM = at_is_int(AT_) ? 16 : 8;
  // MFMAs per 16-byte activation granule: fp8 operands are 8 bytes, so a granule feeds two
  static constexpr int MPG = AT_ == AT_F8 ? 2 : 1;
  static constexpr int NJ = 4 * MPG;
  static constexpr int KL = KPM * NJ;                 // 32 (fp16) / 64 (int8, fp8)
  static constexpr int KS = 4 * KL;                   // 128 / 256: one LDS row is 256 bytes either way
  static constexpr int WL = KL * BITS / 32;           // 32-bit weight words per lane per k-step
  static constexpr int ROW_BYTES = 256;
  static constexpr int LDS_BYTES = (SK_ > 0 ? SK_ : 2) * BM * ROW_BYTES;
  // LDS read prefetch distance in MFMA slots (0: leave the order to the compiler).  Same-box A/B of the
  // 4-wave members, N = K = 4096: int2 x int8 M=512 30.4 -> 26.5 us, M=1024 43.9 -> 35.1 us; uint4 x fp16
  // within +-2 % (M=2048 and 11008 x 4096 M=512 slower): on for the integer members only.
  // 256 x 256 / 8 waves: 4 slots ahead pays for sub-byte integer weights (int2 x int8 4096^3 82.9 -> 79.5 us,
  // 8192^3 530 -> 499 us) and costs 50 % with 8-bit weights (256 registers, no room to schedule)
  static constexpr int PD = (SK_ == 0 && at_is_int(AT_)) ? (NWAVES_ == 4 ? 6 : (T::SUBBYTE ? 4 : 0)) : 0;
};

// ------------------------------------------------------------------------------------------
// extraction order -> natural k order, resolved at compile time (v_perm_b32 per output register;
// nothing at all for the LOP3 layouts, whose extraction order already is the natural one)
// ------------------------------------------------------------------------------------------
template <class T, int LAYOUT>
constexpr int slot_of_elem(int e) {
  for (int x = 0; x < T::EPW; ++x)
    if (T::src_elem(LAYOUT, x) == e) return x;
  return -1;
}

template <class T, int LAYOUT, int I>
__device__ __forceinline__ uint32_t natural_pair_f16(const half2_t (&q)[T::EPW / 2]) {
  constexpr int sa = slot_of_elem<T, LAYOUT>(2 * I), sb = slot_of_elem<T, LAYOUT>(2 * I + 1);
  if constexpr (sa == 2 * I && sb == 2 * I + 1) {
    return as_u32(q[I]);
  } else {
    constexpr uint32_t sel = ((uint32_t)(4 + 2 * (sb % 2) + 1) << 24) | ((uint32_t)(4 + 2 * (sb % 2)) << 16) |
                             ((uint32_t)(2 * (sa % 2) + 1) << 8) | (uint32_t)(2 * (sa % 2));
    return __builtin_amdgcn_perm(as_u32(q[sb / 2]), as_u32(q[sa / 2]), sel);
  }
}
template <class T, int LAYOUT, int... I>
__device__ __forceinline__ void to_natural_f16(const half2_t (&q)[T::EPW / 2], uint32_t (&out)[T::EPW / 2],
                                               std::integer_sequence<int, I...>) {
  ((out[I] = natural_pair_f16<T, LAYOUT, I>(q)), ...);
}

template <class T, int LAYOUT, int I>
__device__ __forceinline__ uint32_t natural_quad_i8(const uint32_t (&q)[T::EPW / 4]) {
  constexpr int s0 = slot_of_elem<T, LAYOUT>(4 * I), s1 = slot_of_elem<T, LAYOUT>(4 * I + 1),
                s2 = slot_of_elem<T, LAYOUT>(4 * I + 2), s3 = slot_of_elem<T, LAYOUT>(4 * I + 3);
  if constexpr (s0 == 4 * I && s1 == 4 * I + 1 && s2 == 4 * I + 2 && s3 == 4 * I + 3) {
    return q[I];
  } else {
    // two bytes at a time: {s0, s1} then {s2, s3}, then merge the halves
    constexpr uint32_t selA = ((uint32_t)(4 + (s1 % 4)) << 8) | (uint32_t)(s0 % 4);
    constexpr uint32_t selB = ((uint32_t)(4 + (s3 % 4)) << 8) | (uint32_t)(s2 % 4);
    const uint32_t lo = __builtin_amdgcn_perm(q[s1 / 4], q[s0 / 4], selA | 0x0C0C0000u);
    const uint32_t hi = __builtin_amdgcn_perm(q[s3 / 4], q[s2 / 4], selB | 0x0C0C0000u);
    return lo | (hi << 16);
  }
}
template <class T, int LAYOUT, int... I>
__device__ __forceinline__ void to_natural_i8(const uint32_t (&q)[T::EPW / 4], uint32_t (&out)[T::EPW / 4],
                                              std::integer_sequence<int, I...>) {
  ((out[I] = natural_quad_i8<T, LAYOUT, I>(q)), ...);
}

// ------------------------------------------------------------------------------------------
// weight words of one lane for one k-step -> NJ MFMA operands (4 x 32-bit each), natural k order
// ------------------------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void dequant_lane_f16(const uint32_t (&w)[P::WL], half_t zf, half2_t s2, half2_t z2,
                                                 const DecodeCtx& cx, const Lut16& lut,
                                                 uint32_t (&frag)[P::NJ][4]) {
  using T = typename P::T;
  constexpr int EPW = P::EPW;
  if constexpr (T::SUBBYTE) {
    // one word holds EPW >= 8 elements = EPW / 8 fragments
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      half2_t q[EPW / 2];
      if constexpr (P::KIND == DK_LUT4) {
        lut16_word(lut, w[wi], q);
      } else {
        F16Unpack<T::BITS>::run(w[wi] ^ (P::KIND == DK_INT1 ? cx.flip : 0u), zf, cx.magic, q);
      }
#pragma unroll
      for (int i = 0; i < EPW / 2; ++i) {
        if constexpr (P::MODE == MD_S || P::MODE == MD_ZQ) q[i] = q[i] * s2;
        if constexpr (P::MODE == MD_ZO) q[i] = (q[i] - z2) * s2;
        if constexpr (P::MODE == MD_ZR) {
          half2_t t = q[i] * s2;
          asm volatile("" : "+v"(t));   // two roundings, no fma contraction
          q[i] = t - z2;
        }
      }
      uint32_t nat[EPW / 2];
      to_natural_f16<T, P::LAYOUT>(q, nat, std::make_integer_sequence<int, EPW / 2>{});
#pragma unroll
      for (int i = 0; i < EPW / 2; ++i) {
        const int e = wi * EPW + 2 * i;   // element index inside the lane's KL
        frag[e / 8][(e % 8) / 2] = nat[i];
      }
    }
  } else {
    // 8-bit (4 per word) and 16-bit (2 per word) weights: already in natural order
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      half2_t q[EPW / 2 > 0 ? EPW / 2 : 1];
      if constexpr (P::KIND == DK_INT8) {
        uint32_t x = w[wi] ^ cx.flip;
        half2_t off = splat(cx.off8 + zf);
        if constexpr (P::MODE == MD_ZQ) {
          // (w - zero) in the int8 storage type (quantization.py:208-217): wraps mod 256, signed byte
          x = sub_bytes_mod(w[wi], (uint32_t)(int)(float)zf * 0x01010101u) ^ 0x80808080u;
          off = splat((half_t)1152.0f);
        }
        q[0] = as_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04010400u)) - off;
        q[1] = as_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04030402u)) - off;
      } else if constexpr (P::KIND == DK_E4M3) {
        half2_t t[2];
        unpack_e4m3_f16<P::STRICT>(w[wi], t);
        q[0] = t[0]; q[1] = t[1];
      } else if constexpr (P::KIND == DK_E5M2) {
        half2_t t[2];
        unpack_e5m2_f16(w[wi], t);
        q[0] = t[0]; q[1] = t[1];
      } else {
        q[0] = as_h2(w[wi]);
      }
#pragma unroll
      for (int i = 0; i < EPW / 2; ++i) {
        if constexpr (P::MODE == MD_S || P::MODE == MD_ZQ) q[i] = q[i] * s2;
        if constexpr (P::MODE == MD_ZO) q[i] = (q[i] - z2) * s2;
        if constexpr (P::MODE == MD_ZR) {
          half2_t t = q[i] * s2;
          asm volatile("" : "+v"(t));
          q[i] = t - z2;
        }
        const int e = wi * EPW + 2 * i;
        frag[e / 8][(e % 8) / 2] = as_u32(q[i]);
      }
    }
  }
}

// bfloat16 flavour: natural order straight away (plain layout), one rounding per element
template <class P>
__device__ __forceinline__ void dequant_lane_bf16(const uint32_t (&w)[P::WL], float zf, float s, bool is_signed,
                                                  uint32_t flip, const Lut16& lut, uint32_t (&frag)[P::NJ][4], float z = 0.f) {
  constexpr int ZM = P::MODE == MD_ZO ? 1 : P::MODE == MD_ZR ? 2 : 0;      // bfloat16 zero-point modes: one more rounding
  using T = typename P::T;
  constexpr int EPW = P::EPW;
  constexpr bool SC = P::MODE != MD_NONE;
  if constexpr (P::KIND == DK_LUT4) {
    // nf4 / fp4: table entries are bfloat16 bit patterns; scale with one rounding, then natural order
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      half2_t q[EPW / 2];
      lut16_word(lut, w[wi], q);
      if constexpr (SC) {
#pragma unroll
        for (int i = 0; i < EPW / 2; ++i) q[i] = as_h2(bf16x2_scale(as_u32(q[i]), s));
      }
      uint32_t nat[EPW / 2];
      to_natural_f16<T, P::LAYOUT>(q, nat, std::make_integer_sequence<int, EPW / 2>{});
#pragma unroll
      for (int i = 0; i < EPW / 2; ++i) {
        const int e = wi * EPW + 2 * i;
        frag[e / 8][(e % 8) / 2] = nat[i];
      }
    }
  } else if constexpr (P::KIND == DK_E4M3) {
    // exact e4m3 -> fp16 -> float, then the scale and one bfloat16 rounding
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      half2_t t[2];
      unpack_e4m3_f16<false>(w[wi], t);
      const float sc = SC ? s : 1.f;
      const int e0 = wi * 4;
      frag[e0 / 8][(e0 % 8) / 2] = cvt_pk_bf16((float)t[0][0] * sc, (float)t[0][1] * sc);
      frag[e0 / 8][(e0 % 8) / 2 + 1] = cvt_pk_bf16((float)t[1][0] * sc, (float)t[1][1] * sc);
    }
  } else if constexpr (T::SUBBYTE) {
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      uint32_t pk[EPW / 2];
      unpack_word_bf16<T::BITS, 1, ZM>(w[wi] ^ (P::KIND == DK_INT1 ? flip : 0u), zf, s, SC, pk, z);
#pragma unroll
      for (int i = 0; i < EPW / 2; ++i) {
        const int e = wi * EPW + 2 * i;
        frag[e / 8][(e % 8) / 2] = pk[i];
      }
    }
  } else if constexpr (P::KIND == DK_INT8) {
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b8 = (int)((w[wi] >> (8 * e)) & 0xFFu);
        if constexpr (P::MODE == MD_ZQ) v[e] = (float)(int)(int8_t)(b8 - (int)zf);   // int8 storage arithmetic wraps
        else v[e] = (float)(is_signed ? (int)(int8_t)b8 : b8);
      }
      dequant_pair_bf16<ZM>(v[0], v[1], 0.f, s, z, SC);
      dequant_pair_bf16<ZM>(v[2], v[3], 0.f, s, z, SC);
      const int e0 = wi * 4;
      frag[e0 / 8][(e0 % 8) / 2] = cvt_pk_bf16(v[0], v[1]);
      frag[e0 / 8][(e0 % 8) / 2 + 1] = cvt_pk_bf16(v[2], v[3]);
    }
  } else {   // native bf16 weights: 2 per word
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) frag[(wi * 2) / 8][((wi * 2) % 8) / 2] = w[wi];
  }
}

template <class P>
__device__ __forceinline__ void dequant_lane_i8(const uint32_t (&w)[P::WL], uint32_t zp4, uint32_t flip,
                                                uint32_t (&frag)[P::NJ][4]) {
  using T = typename P::T;
  constexpr int EPW = P::EPW;
  if constexpr (T::SUBBYTE) {
    constexpr int NQ = I8Unpack<T::BITS>::NQUAD;   // == EPW / 4
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) {
      uint32_t t[NQ];
      I8Unpack<T::BITS>::run(w[wi] ^ flip, t);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        if constexpr (T::BITS <= 2) t[i] = sub_bytes_tbl(t[i], sub_bytes(0x03020100u, zp4));   // one byte permute (the table is loop invariant)
        else t[i] = sub_bytes(t[i], zp4);
      }
      uint32_t nat[NQ];
      to_natural_i8<T, P::LAYOUT>(t, nat, std::make_integer_sequence<int, NQ>{});
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = wi * EPW + 4 * i;
        frag[e / 16][(e % 16) / 4] = nat[i];
      }
    }
  } else {
#pragma unroll
    for (int wi = 0; wi < P::WL; ++wi) frag[wi / 4][wi % 4] = w[wi];
  }
}

template <int NW_, bool NT = false>   // NT: non-temporal (weights one CU reads once: landed ~18 % sooner, MI355X_MICROARCH "nt-weights")
__device__ __forceinline__ void load_lane_words(const uint8_t* p, uint32_t (&w)[NW_]) {
  if constexpr (NW_ % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NW_ / 4; ++q) {
      const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + q) : reinterpret_cast<const u32x4*>(p)[q];
      w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
    }
  } else if constexpr (NW_ == 2) {
    const u32x2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)) : *reinterpret_cast<const u32x2*>(p);
    w[0] = v[0]; w[1] = v[1];
  } else {
    static_assert(NW_ == 1, "unsupported lane word count");
    w[0] = NT ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)) : *reinterpret_cast<const uint32_t*>(p);
  }
}

// ------------------------------------------------------------------------------------------
// output of one lane's 4 consecutive columns [nb, nb+4) of row m: cast to out_dtype, then + bias
// ------------------------------------------------------------------------------------------
template <class P, class ACC>
__device__ __forceinline__ void store_quad(const GemmArgs& a, const ACC v, int m, int nb) {
  constexpr bool F16 = P::AT == AT_F16, F8 = P::AT == AT_F8, FACC = F16 || F8;
  float bias_f[4] = {0.f, 0.f, 0.f, 0.f};
  int bias_i[4] = {0, 0, 0, 0};
  if (a.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (F16 && P::BF) bias_f[i] = bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[nb + i]);
      else if constexpr (F16) bias_f[i] = (float)reinterpret_cast<const half_t*>(a.bias)[nb + i];
      else if constexpr (F8) bias_f[i] = 0.f;   // the reference defines no fp8 bias operand
      else if (!a.epi_row) bias_i[i] = (int)reinterpret_cast<const int8_t*>(a.bias)[nb + i];
    }
  }
  const long base = (long)m * a.N + nb;
  if constexpr (FACC) {
    if (a.out_dtype == WQAA_F16) {
      half_t h[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h[i] = (half_t)v[i];
        if (a.has_bias) h[i] = h[i] + (half_t)bias_f[i];
      }
      const half2_t lo = {h[0], h[1]}, hi = {h[2], h[3]};
      *reinterpret_cast<u32x2*>(reinterpret_cast<half_t*>(a.C) + base) = u32x2{as_u32(lo), as_u32(hi)};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) store_out(a.C, base + i, v[i], a.out_dtype, a.has_bias != 0, bias_f[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a.epi_row) store_out_fused(a.C, base + i, v[i], a.epi_row[m], a.epi_tensor, a.has_bias != 0, a.bias, nb + i);
      else store_out(a.C, base + i, v[i], a.out_dtype, a.has_bias != 0, bias_i[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
template <class P>
struct BLane {
  uint32_t w[P::NFW][P::WL];
  uint32_t s[P::NFW];
  uint32_t z[P::NFW];
};

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_kernel(const GemmArgs a) {
  using T = typename P::T;
  constexpr int MF = P::MF, NFW = P::NFW, NJ = P::NJ, WL = P::WL, MODE = P::MODE;
  constexpr bool F16 = P::AT == AT_F16;
  constexpr bool F8 = P::AT == AT_F8;
  constexpr bool FACC = F16 || F8;                        // fp32 accumulators
  constexpr int ASZ = F16 ? 2 : 1;                        // bytes per activation element (in LDS)
  constexpr bool A4 = P::AT == AT_I4;                     // packed int4 activations in memory
  using acc_t = typename std::conditional<FACC, f32x4, i32x4>::type;
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;
  constexpr int ZPB = 8 / ZB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  WQ_TRACE_DECL;
  WQ_TRACE(0);
  const int fr = lane & 15;          // fragment row (weight n / activation m)
  const int kb = lane >> 4;          // k-block of the lane

  const TileOfBlock tob = tile_of_block(a, (int)blockIdx.x, (int)gridDim.x);
  const int split = tob.split, tile_m = tob.tile_m, tile_n = tob.tile_n;
  const int m0 = tile_m * P::BM;
  const int n0 = tile_n * P::BN + wave * (NFW * 16);

  const uint8_t* Ap = reinterpret_cast<const uint8_t*>(a.A);
  const uint8_t* Bp = reinterpret_cast<const uint8_t*>(a.B);
  const uint16_t* Sp = reinterpret_cast<const uint16_t*>(a.scale);
  const uint16_t* Zp = reinterpret_cast<const uint16_t*>(a.zeros);
  const uint8_t* Qp = reinterpret_cast<const uint8_t*>(a.zeros);

  // ---- activation staging: AG granules (16 B) per thread per k-step ----
  // 16 consecutive threads cover one 256-byte row.  Inside a row the two 8-lane halves (the
  // ds_write_b128 service groups, 128-byte bank window) take natural granules {0,1,4,5,8,9,12,13} and
  // {2,3,6,7,...}: their physical slots (j*4+kb)^r are then distinct mod 8 - no write conflicts.
  constexpr int AG = P::AG;
  u32x4 areg[AG];
  // addresses cost registers here: a wave-uniform base (SGPRs) + one 32-bit byte offset per granule in
  // memory, and ONE LDS offset - rows of successive granules are THREADS / 16 apart, a multiple of 16,
  // so the swizzle term is the same and the rest is a compile-time stride
  const long a_row_bytes = A4 ? (long)(a.K / 2) : (long)a.K * ASZ;
  const uint8_t* a_tile = Ap + (long)m0 * a_row_bytes;
  uint32_t aoff[AG];
  constexpr int LDS_IT_STRIDE = (P::THREADS / 16) * P::ROW_BYTES;
  int a_lds_off0;
  {
    const int r0 = tid >> 4, q = tid & 15;
    const int ns = ((q & 7) >> 1) * 4 + (q & 1) + ((q >> 3) << 1);   // natural granule = kb' * 4 + j'
    const int phys = (((ns & 3) << 2) | (ns >> 2)) ^ (r0 & 15);
    a_lds_off0 = r0 * P::ROW_BYTES + phys * 16;
#pragma unroll
    for (int it = 0; it < AG; ++it) {
      int r = it * (P::THREADS / 16) + r0;
      r = m0 + r < a.M ? r : a.M - 1 - m0;                         // clamped rows are never stored
      aoff[it] = (uint32_t)(r * (int)a_row_bytes) + (uint32_t)(ns * (A4 ? 8 : 16));
    }
  }
  // packed int4 activations: a granule of 16 elements is 8 bytes in memory; widened to int8 on the way
  // into LDS, so everything downstream is the int8 path
  auto granule_load = [&](const uint8_t* p) -> u32x4 {
    if constexpr (A4) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(p);
      return u32x4{v[0], v[1], 0u, 0u};
    } else {
      return *reinterpret_cast<const u32x4*>(p);
    }
  };
  auto granule_lds = [&](const u32x4 v) -> u32x4 {
    if constexpr (A4) {
      u32x4 o;
      uint32_t x0, x1, x2, x3;
      widen_nibbles(v[0], x0, x1);
      widen_nibbles(v[1], x2, x3);
      o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
      return o;
    } else {
      return v;
    }
  };
  constexpr int ASTEP = A4 ? P::KS / 2 : P::KS * ASZ;     // bytes of one k-step in a row of A
  auto a_load = [&](int t) {
    const long koff = (long)t * ASTEP;
#pragma unroll
    for (int it = 0; it < AG; ++it) areg[it] = granule_load(a_tile + koff + aoff[it]);
  };
  auto a_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < AG; ++it)
      *reinterpret_cast<u32x4*>(smem_raw + buf * (P::BM * P::ROW_BYTES) + a_lds_off0 + it * LDS_IT_STRIDE) = granule_lds(areg[it]);
  };

  // ---- weight lane loads ----
  int nrow[NFW];
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    const int n = n0 + nf * 16 + fr;
    nrow[nf] = n < a.N ? n : a.N - 1;
  }
  const uint8_t* bptr[NFW];
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) bptr[nf] = Bp + (long)nrow[nf] * a.row_bytes + (long)kb * (WL * 4);
  auto b_load = [&](int t, BLane<P>& b) {
    const int kidx = t * 4 + kb;    // index of the lane's KL-wide k-block
    int gi = 0;
    if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (kidx >> a.gq_shift) : (int)__umulhi((uint32_t)kidx, a.gq_magic);
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      load_lane_words<WL>(bptr[nf] + (long)t * (4 * WL * 4), b.w[nf]);
      if constexpr (MODE != MD_NONE) b.s[nf] = Sp[(long)nrow[nf] * a.kg + gi];
      if constexpr (MODE == MD_ZO || MODE == MD_ZR) b.z[nf] = Zp[(long)nrow[nf] * a.kg + gi];
      if constexpr (MODE == MD_ZQ) b.z[nf] = Qp[(long)gi * a.zq_row_bytes + nrow[nf] / ZPB];
    }
  };

  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
  cx.flip = 0u;
  if (P::KIND == DK_INT1 && a.is_signed) cx.flip = 0xFFFFFFFFu;
  if (P::KIND == DK_INT8 && a.is_signed) cx.flip = 0x80808080u;
  if (A4 && P::KIND == DK_INT4 && a.is_signed) cx.flip = 0x88888888u;   // two's-complement weight nibbles
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  const uint32_t zp4 = (!F16 && a.is_signed && T::SUBBYTE) ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) {
      lut = make_fp4_lut(P::BF);
    } else {
      lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
    }
  }

  acc_t acc[MF][NFW];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = acc_t{0, 0, 0, 0};

  // one k-step of this wave: dequantise its NFW weight fragments, then MF x NFW x NJ MFMAs against the
  // activation tile at `abuf`
  auto compute_step = [&](const BLane<P>& bl, const unsigned char* abuf) {
    // dequantise this wave's weight fragments for the whole k-step
    uint32_t bfrag[NFW][NJ][4];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      if constexpr (F16) {
        half_t zf = cx.zf;
        if constexpr (MODE == MD_ZQ) {
          const uint32_t zq = (bl.z[nf] >> ((nrow[nf] % ZPB) * ZB)) & ((1u << ZB) - 1u);
          zf = (half_t)(float)zq;
        }
        const half2_t s2 = MODE != MD_NONE ? splat(bits_to_half(bl.s[nf])) : splat((half_t)1.0f);
        const half2_t z2 = (MODE == MD_ZO || MODE == MD_ZR) ? splat(bits_to_half(bl.z[nf])) : splat((half_t)0.0f);
        if constexpr (P::BF)
          dequant_lane_bf16<P>(bl.w[nf], (float)zf, MODE != MD_NONE ? bf16_bits_to_float(bl.s[nf]) : 1.f, a.is_signed != 0,
                               cx.flip, lut, bfrag[nf], (MODE == MD_ZO || MODE == MD_ZR) ? bf16_bits_to_float(bl.z[nf]) : 0.f);
        else
          dequant_lane_f16<P>(bl.w[nf], zf, s2, z2, cx, lut, bfrag[nf]);
      } else if constexpr (F8) {
        // fp8 weights are MFMA operands as stored: fragment j = the lane's bytes [8j, 8j+8)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          bfrag[nf][j][0] = bl.w[nf][2 * j];
          bfrag[nf][j][1] = bl.w[nf][2 * j + 1];
        }
      } else {
        dequant_lane_i8<P>(bl.w[nf], zp4, cx.flip, bfrag[nf]);
      }
    }

    if constexpr (F8) {
      // dense fp8: the 128-deep scaled MFMA with unit scales (E8M0 127 = 2^0).  A lane feeds 32
      // consecutive bytes of its 64-byte k-block per instruction - two activation granules against 8
      // weight words - so a k-step is 2 instructions per (mf, nf) instead of 8 of the 32-deep form,
      // whose measured ceiling on this chip equals the fp16 one (MI355X_MICROARCH: 2.05 vs 4.66 PF).
      constexpr int WFMT = P::KIND == DK_E5M2 ? 1 : 0, AFMT = (P::FLAGS & FL_ABF8) ? 1 : 0;   // 0 e4m3, 1 e5m2
      typedef int i32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const unsigned char* rowp = abuf + (mf * 16 + fr) * P::ROW_BYTES;
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(rowp + ((((2 * hp) << 2) | kb) ^ fr) * 16);
          const u32x4 a1 = *reinterpret_cast<const u32x4*>(rowp + ((((2 * hp + 1) << 2) | kb) ^ fr) * 16);
          const i32x8 av8 = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf) {
            i32x8 wv8;
#pragma unroll
            for (int e = 0; e < 8; ++e) wv8[e] = (int)bl.w[nf][8 * hp + e];
            acc[mf][nf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv8, av8, acc[mf][nf], WFMT, AFMT, 0, 0x7F7F7F7F, 0,
                                                                          0x7F7F7F7F);
          }
        }
      }
    } else if constexpr (P::PD > 0) {
      // 4-wave integer members: left alone the scheduler sinks every ds_read next to its use (two reads,
      // wait, four MFMAs) and nothing else on the SIMD hides that LDS latency.  Slot s = (gq, mf)
      // reads one activation granule and feeds NFW MFMAs; the reads run PD slots ahead through a
      // register ring, and the fences pin MFMA / LDS / global-memory order to the source order while
      // VALU + SALU (the weight decode) float between the MFMAs.
      constexpr int SLOTS = 4 * MF;
      constexpr int PD = P::PD < SLOTS ? P::PD : SLOTS;
      const unsigned char* arow = abuf + fr * P::ROW_BYTES;
      auto lds_read = [&](int sl) -> u32x4 {
        const int gq = sl / MF, mf = sl % MF;
        return *reinterpret_cast<const u32x4*>(arow + mf * (16 * P::ROW_BYTES) + ((((gq << 2) | kb) ^ fr) * 16));
      };
      u32x4 ring[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) ring[i] = lds_read(i);
      __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) {
        const int gq = sl / MF, mf = sl % MF;
        const u32x4 av = ring[sl % PD];
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          const u32x4 bv = {bfrag[nf][gq][0], bfrag[nf][gq][1], bfrag[nf][gq][2], bfrag[nf][gq][3]};
          if constexpr (F16) {
            if constexpr (P::BF)
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv),
                                                                   __builtin_bit_cast(bf16x8_t, av), acc[mf][nf], 0, 0, 0);
            else
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv),
                                                                  __builtin_bit_cast(half8_t, av), acc[mf][nf], 0, 0, 0);
          } else {
            acc[mf][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bv),
                                                               __builtin_bit_cast(i32x4, av), acc[mf][nf], 0, 0, 0);
          }
        }
        if (sl + PD < SLOTS) ring[sl % PD] = lds_read(sl + PD);
        __builtin_amdgcn_sched_barrier(0x6);
      }
    } else {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {          // activation granule (kb, gq) of the lane's k-block
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int phys = ((gq << 2) | kb) ^ fr;
        const u32x4 av = *reinterpret_cast<const u32x4*>(abuf + (mf * 16 + fr) * P::ROW_BYTES + phys * 16);
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          if constexpr (F16) {
            const u32x4 bv = {bfrag[nf][gq][0], bfrag[nf][gq][1], bfrag[nf][gq][2], bfrag[nf][gq][3]};
            if constexpr (P::BF)
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv),
                                                                   __builtin_bit_cast(bf16x8_t, av), acc[mf][nf], 0, 0, 0);
            else
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv),
                                                                  __builtin_bit_cast(half8_t, av), acc[mf][nf], 0, 0, 0);
          } else if constexpr (F8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const u32x2 b2 = {bfrag[nf][2 * gq + h][0], bfrag[nf][2 * gq + h][1]};
              const u32x2 a2 = {av[2 * h], av[2 * h + 1]};
              const long bl = __builtin_bit_cast(long, b2), al = __builtin_bit_cast(long, a2);
              constexpr bool WB = P::KIND == DK_E5M2, AB = (P::FLAGS & FL_ABF8) != 0;
              if constexpr (!WB && !AB) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bl, al, acc[mf][nf], 0, 0, 0);
              if constexpr (!WB && AB) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(bl, al, acc[mf][nf], 0, 0, 0);
              if constexpr (WB && !AB) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(bl, al, acc[mf][nf], 0, 0, 0);
              if constexpr (WB && AB) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(bl, al, acc[mf][nf], 0, 0, 0);
            }
          } else {
            const u32x4 bv = {bfrag[nf][gq][0], bfrag[nf][gq][1], bfrag[nf][gq][2], bfrag[nf][gq][3]};
            acc[mf][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bv),
                                                               __builtin_bit_cast(i32x4, av), acc[mf][nf], 0, 0, 0);
          }
        }
      }
    }
    }

  };

  if constexpr (P::SK > 0) {
    constexpr int S = P::SK;
    const int t0 = split * S;
    u32x4 areg_s[S][AG];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int t = t0 + q < a.nsteps ? t0 + q : a.nsteps - 1;
      const long koff = (long)t * ASTEP;
#pragma unroll
      for (int it = 0; it < AG; ++it) areg_s[q][it] = granule_load(a_tile + koff + aoff[it]);
    }
    BLane<P> bs[S];
    // Scale / Zeros: with one group per k-step (g = 128 for fp16) the S = 4 steps of this workgroup use 4
    // consecutive groups - ONE 8-byte load per row instead of four 2-byte loads.  A 2-byte load per lane
    // touches 16 cache lines per wave instruction (16 rows), exactly like the 16-byte weight load, so
    // the per-step form spends two thirds of its memory transactions on 64 bytes of metadata.
    constexpr bool WIDE_OK = S == 4 && (MODE == MD_S || MODE == MD_ZO || MODE == MD_ZR);
    const bool wide = WIDE_OK && a.gq_shift == 2 && (a.kg & 3) == 0 && t0 + S <= a.nsteps;   // wave-uniform
    // groups of 512 and more (and per-channel scales): the four steps share ONE group - one 2-byte load
    const bool uni = WIDE_OK && a.gq_shift >= 4 && t0 + S <= a.nsteps;
    if (uni) {
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) load_lane_words<WL, true>(bptr[nf] + (long)(t0 + q) * (4 * WL * 4), bs[q].w[nf]);
      const int gi = (t0 * 4) >> a.gq_shift;
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const uint32_t sv = Sp[(long)nrow[nf] * a.kg + gi];
        uint32_t zv = 0;
        if constexpr (MODE == MD_ZO || MODE == MD_ZR) zv = Zp[(long)nrow[nf] * a.kg + gi];
#pragma unroll
        for (int q = 0; q < S; ++q) { bs[q].s[nf] = sv; bs[q].z[nf] = zv; }
      }
    } else if (WIDE_OK && a.gq_shift == 3 && (a.kg & 1) == 0 && t0 + S <= a.nsteps) {
      // g = 256: two groups per four steps - one 4-byte load
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) load_lane_words<WL, true>(bptr[nf] + (long)(t0 + q) * (4 * WL * 4), bs[q].w[nf]);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const uint32_t sv = *reinterpret_cast<const uint32_t*>(Sp + (long)nrow[nf] * a.kg + (t0 >> 1));
        uint32_t zv = 0;
        if constexpr (MODE == MD_ZO || MODE == MD_ZR) zv = *reinterpret_cast<const uint32_t*>(Zp + (long)nrow[nf] * a.kg + (t0 >> 1));
#pragma unroll
        for (int q = 0; q < S; ++q) {
          bs[q].s[nf] = (sv >> (16 * (q >> 1))) & 0xFFFFu;
          bs[q].z[nf] = (zv >> (16 * (q >> 1))) & 0xFFFFu;
        }
      }
    } else if (wide) {
#pragma unroll
      for (int q = 0; q < S; ++q)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) load_lane_words<WL, true>(bptr[nf] + (long)(t0 + q) * (4 * WL * 4), bs[q].w[nf]);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const u32x2 sv = *reinterpret_cast<const u32x2*>(Sp + (long)nrow[nf] * a.kg + t0);
        u32x2 zv = {0u, 0u};
        if constexpr (MODE == MD_ZO || MODE == MD_ZR) zv = *reinterpret_cast<const u32x2*>(Zp + (long)nrow[nf] * a.kg + t0);
#pragma unroll
        for (int q = 0; q < S; ++q) {
          bs[q].s[nf] = (sv[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
          bs[q].z[nf] = (zv[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < S; ++q) b_load(t0 + q < a.nsteps ? t0 + q : a.nsteps - 1, bs[q]);
    }
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int it = 0; it < AG; ++it)
        *reinterpret_cast<u32x4*>(smem_raw + q * (P::BM * P::ROW_BYTES) + a_lds_off0 + it * LDS_IT_STRIDE) = granule_lds(areg_s[q][it]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < S; ++q)
      if (t0 + q < a.nsteps) compute_step(bs[q], smem_raw + q * (P::BM * P::ROW_BYTES));
  } else {
    const int t_begin = udiv_magic(split * a.nsteps, a.ksplit, a.mg_ksplit);
    const int nsteps = udiv_magic((split + 1) * a.nsteps, a.ksplit, a.mg_ksplit);   // end of this workgroup's k range
    BLane<P> bcur, bnext;
    if constexpr (P::WIDE) {
      // one group per k-step: the groups of four consecutive steps come with ONE 8-byte load per row, a
      // block ahead; the loop is unrolled by four so each step's half is a compile-time pick
      u32x2 gs_cur[NFW], gz_cur[NFW], gs_nxt[NFW], gz_nxt[NFW];
      auto g_load = [&](int first, u32x2 (&gs)[NFW], u32x2 (&gz)[NFW]) {
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          gs[nf] = *reinterpret_cast<const u32x2*>(Sp + (long)nrow[nf] * a.kg + first);
          if constexpr (MODE == MD_ZO || MODE == MD_ZR) gz[nf] = *reinterpret_cast<const u32x2*>(Zp + (long)nrow[nf] * a.kg + first);
          else gz[nf] = u32x2{0u, 0u};
        }
      };
      auto w_load = [&](int t, BLane<P>& b) {
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) load_lane_words<WL>(bptr[nf] + (long)t * (4 * WL * 4), b.w[nf]);
      };
      const int tb4 = t_begin & ~3;               // blocks are aligned to 4 steps (the host keeps split ranges aligned)
      a_load(t_begin);
      w_load(t_begin, bcur);
      g_load(tb4, gs_cur, gz_cur);
      WQ_TRACE(1);
      a_store(t_begin & 1);
      __syncthreads();
      WQ_TRACE(2);
      for (int t4 = tb4; t4 < nsteps; t4 += 4) {
        const int nb = t4 + 4 < nsteps ? t4 + 4 : t4;      // next block (the last one reloads itself)
        g_load(nb, gs_nxt, gz_nxt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t4 + q;
          if (t < t_begin || t >= nsteps) continue;         // wave-uniform
          const int tn = t + 1 < nsteps ? t + 1 : t;
          a_load(tn);
          w_load(tn, bnext);
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf) {
            bcur.s[nf] = (gs_cur[nf][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
            bcur.z[nf] = (gz_cur[nf][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
          }
          compute_step(bcur, smem_raw + (t & 1) * (P::BM * P::ROW_BYTES));
          a_store((t + 1) & 1);
          __syncthreads();
          WQ_TRACE_IF(t == t_begin, 3);
          bcur = bnext;
        }
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) { gs_cur[nf] = gs_nxt[nf]; gz_cur[nf] = gz_nxt[nf]; }
      }
    } else {
    a_load(t_begin);
    b_load(t_begin, bcur);
    a_store(t_begin & 1);
    __syncthreads();

    for (int t = t_begin; t < nsteps; ++t) {
      const int tn = t + 1 < nsteps ? t + 1 : t;    // last step reloads itself: loads stay unconditional
      a_load(tn);
      b_load(tn, bnext);
      compute_step(bcur, smem_raw + (t & 1) * (P::BM * P::ROW_BYTES));
      a_store((t + 1) & 1);
      __syncthreads();
      bcur = bnext;
    }
    }
  }

  // ---- epilogue: D[i][col]: weight row n = nbase + kb * 4 + i, activation row m = mbase + fr ----
  WQ_TRACE(4);
  if (a.ksplit > 1) {
    acc_t* ws = reinterpret_cast<acc_t*>(a.ws);
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      const int nb = n0 + nf * 16 + kb * 4;
      if (nb >= a.N) continue;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int m = m0 + mf * 16 + fr;
        if (m >= a.M) continue;
        acc_t* dst = ws + ((((long)split * a.M + m) * a.N + nb) >> 2);
        // the partial sums are read once, by another kernel: left dirty in L2 they are written back at the kernel boundary
        // (MI355X_MICROARCH "boundary": + B / 6 TB/s behind B dirty bytes)
        // (same-process A/B, profiles/r03_ab_ws_policy.txt: 4096^2 M = 64 14.6 -> 12.2 us, M = 128 19.5 -> 17.1, M = 256 26.1 -> 22.4)
        if ((a.ws_policy & 3) == 1) __builtin_nontemporal_store(acc[mf][nf], dst);
        else if ((a.ws_policy & 3) == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[mf][nf]) : "memory");
        else if ((a.ws_policy & 3) == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[mf][nf]) : "memory");
        else *dst = acc[mf][nf];
      }
    }
    WQ_TRACE(5);
    WQ_TRACE_DUMP(P::NWAVES);
    return;
  }
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    const int nb = n0 + nf * 16 + kb * 4;
    if (nb >= a.N) continue;
    float bias_f[4] = {0.f, 0.f, 0.f, 0.f};
    int bias_i[4] = {0, 0, 0, 0};
    if (a.has_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (F16 && P::BF) bias_f[i] = bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[nb + i]);
        else if constexpr (F16) bias_f[i] = (float)reinterpret_cast<const half_t*>(a.bias)[nb + i];
        else if constexpr (F8) bias_f[i] = 0.f;   // the reference defines no fp8 bias operand
        else if (!a.epi_row) bias_i[i] = (int)reinterpret_cast<const int8_t*>(a.bias)[nb + i];
      }
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = m0 + mf * 16 + fr;
      if (m >= a.M) continue;
      const long base = (long)m * a.N + nb;
      if constexpr (FACC) {
        if (a.out_dtype == WQAA_F16) {
          half_t v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = (half_t)acc[mf][nf][i];
            if (a.has_bias) v[i] = v[i] + (half_t)bias_f[i];
          }
          const half2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
          u32x2* dst = reinterpret_cast<u32x2*>(reinterpret_cast<half_t*>(a.C) + base);
          const u32x2 x = {as_u32(lo), as_u32(hi)};
          // a large output leaves the chip write-through (GemmArgs::ws_policy bit 4): nothing dirty for the kernel boundary to write back
          if (a.ws_policy & 16) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(x) : "memory");
          else *dst = x;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) store_out(a.C, base + i, acc[mf][nf][i], a.out_dtype, a.has_bias != 0, bias_f[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (a.epi_row) store_out_fused(a.C, base + i, acc[mf][nf][i], a.epi_row[m], a.epi_tensor, a.has_bias != 0, a.bias, nb + i);
          else store_out(a.C, base + i, acc[mf][nf][i], a.out_dtype, a.has_bias != 0, bias_i[i]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// decode-batch member (M <= 16; the selector uses it for M = 5..8): ONE launch, no partial sums in memory.
// A workgroup owns one 16-row weight fragment and ALL of K; its NW waves take the k-steps round-robin
// and meet in LDS at the end (fixed summation order: deterministic).  Nothing is shared between waves
// inside the loop - no LDS staging, no barrier: the lane reads its activation granules (64 contiguous
// bytes per row and k-step, L2-resident) straight into MFMA operands.  PF k-steps of BOTH streams are
// in flight per wave; for K = 4096 that is the wave's whole share, issued before anything is consumed:
// one memory round trip per launch, which is what a latency-bound decode step wants.
// ------------------------------------------------------------------------------------------
template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_decode_kernel(const GemmArgs a) {
  using T = typename P::T;
  constexpr int MF = P::MF, NJ = P::NJ, WL = P::WL, MODE = P::MODE, NW = P::NWAVES;
  constexpr bool F16 = P::AT == AT_F16, F8 = P::AT == AT_F8, FACC = F16 || F8, A4 = P::AT == AT_I4;
  constexpr int ASZ = F16 ? 2 : 1;
  using acc_t = typename std::conditional<FACC, f32x4, i32x4>::type;
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;
  constexpr int ZPB = 8 / ZB;
  constexpr int PF = 4;                       // k-steps in flight per wave
  static_assert(P::NFW == 1 && MF <= 2, "decode member: one weight fragment per workgroup, M <= 32");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kb = lane >> 4;
  // XCD-aware order: an XCD (block b runs on XCD b % 8) takes a contiguous band of weight rows
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int n0 = blk * 16;
  int nrow = n0 + fr;
  nrow = nrow < a.N ? nrow : a.N - 1;

  const uint8_t* Ap = reinterpret_cast<const uint8_t*>(a.A);
  const uint8_t* Bp = reinterpret_cast<const uint8_t*>(a.B);
  const uint16_t* Sp = reinterpret_cast<const uint16_t*>(a.scale);
  const uint16_t* Zp = reinterpret_cast<const uint16_t*>(a.zeros);
  const uint8_t* Qp = reinterpret_cast<const uint8_t*>(a.zeros);
  const uint8_t* brow = Bp + (long)nrow * a.row_bytes + (long)kb * (WL * 4);
  const long srow = (long)nrow * a.kg;

  // the lane's activation rows: fragment mf -> row mf * 16 + fr (clamped: rows >= M are never stored)
  constexpr int ALB = A4 ? 32 : 64;           // bytes of one lane k-block of A in memory
  const uint8_t* arow[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int m = mf * 16 + fr;
    m = m < a.M ? m : a.M - 1;
    arow[mf] = A4 ? Ap + (long)m * (a.K / 2) + kb * ALB : Ap + (long)m * a.K * ASZ + kb * ALB;
  }

  struct Step {
    BLane<P> b;
    u32x4 araw[MF][A4 ? 2 : 4];
  };
  // Scale / Zeros: a 2-byte load per lane touches 16 cache lines per wave instruction, as many as the
  // 16-byte weight load.  With one group per k-step (g = 128 for fp16) four consecutive steps use four
  // consecutive groups: ONE 8-byte load per row and block of 4 steps (wide = true) instead of four.
  constexpr bool WIDE_OK = MODE == MD_S || MODE == MD_ZO || MODE == MD_ZR;
  const bool wide = WIDE_OK && a.gq_shift == 2 && (a.kg & 3) == 0;   // wave-uniform
  auto load_step = [&](int t, Step& st) {
    const int kidx = t * 4 + kb;
    int gi = 0;
    if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (kidx >> a.gq_shift) : (int)__umulhi((uint32_t)kidx, a.gq_magic);
    load_lane_words<WL, true>(brow + (long)t * (4 * WL * 4), st.b.w[0]);
    if (!wide) {
      if constexpr (MODE != MD_NONE) st.b.s[0] = Sp[srow + gi];
      if constexpr (MODE == MD_ZO || MODE == MD_ZR) st.b.z[0] = Zp[srow + gi];
    }
    if constexpr (MODE == MD_ZQ) st.b.z[0] = Qp[(long)gi * a.zq_row_bytes + nrow / ZPB];
    const long koff = (long)t * (4 * ALB);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int g = 0; g < (A4 ? 2 : 4); ++g) st.araw[mf][g] = *reinterpret_cast<const u32x4*>(arow[mf] + koff + g * 16);
  };

  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
  cx.flip = 0u;
  if (P::KIND == DK_INT1 && a.is_signed) cx.flip = 0xFFFFFFFFu;
  if (P::KIND == DK_INT8 && a.is_signed) cx.flip = 0x80808080u;
  if (A4 && P::KIND == DK_INT4 && a.is_signed) cx.flip = 0x88888888u;
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  const uint32_t zp4 = (!F16 && a.is_signed && T::SUBBYTE) ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) lut = make_fp4_lut(P::BF);
    else lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
  }

  acc_t acc[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) acc[mf] = acc_t{0, 0, 0, 0};

  auto compute = [&](const Step& st) {
    const BLane<P>& bl = st.b;
    uint32_t bfrag[NJ][4];
    if constexpr (F16) {
      half_t zf = cx.zf;
      if constexpr (MODE == MD_ZQ) {
        const uint32_t zq = (bl.z[0] >> ((nrow % ZPB) * ZB)) & ((1u << ZB) - 1u);
        zf = (half_t)(float)zq;
      }
      const half2_t s2 = MODE != MD_NONE ? splat(bits_to_half(bl.s[0])) : splat((half_t)1.0f);
      const half2_t z2 = (MODE == MD_ZO || MODE == MD_ZR) ? splat(bits_to_half(bl.z[0])) : splat((half_t)0.0f);
      if constexpr (P::BF)
        dequant_lane_bf16<P>(bl.w[0], (float)zf, MODE != MD_NONE ? bf16_bits_to_float(bl.s[0]) : 1.f, a.is_signed != 0, cx.flip, lut, bfrag,
                             (MODE == MD_ZO || MODE == MD_ZR) ? bf16_bits_to_float(bl.z[0]) : 0.f);
      else
        dequant_lane_f16<P>(bl.w[0], zf, s2, z2, cx, lut, bfrag);
    } else if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bfrag[j][0] = bl.w[0][2 * j];
        bfrag[j][1] = bl.w[0][2 * j + 1];
      }
    } else {
      dequant_lane_i8<P>(bl.w[0], zp4, cx.flip, bfrag);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        u32x4 v;
        if constexpr (A4) {   // granule g = 8 packed bytes -> 16 int8
          const u32x4 r = st.araw[mf][g >> 1];
          uint32_t x0, x1, x2, x3;
          widen_nibbles(r[2 * (g & 1)], x0, x1);
          widen_nibbles(r[2 * (g & 1) + 1], x2, x3);
          v = u32x4{x0, x1, x2, x3};
        } else {
          v = st.araw[mf][g];
        }
        if constexpr (F16) {
          const u32x4 bv = {bfrag[g][0], bfrag[g][1], bfrag[g][2], bfrag[g][3]};
          if constexpr (P::BF)
            acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, v), acc[mf], 0, 0, 0);
          else
            acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv), __builtin_bit_cast(half8_t, v), acc[mf], 0, 0, 0);
        } else if constexpr (F8) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x2 b2 = {bfrag[2 * g + h][0], bfrag[2 * g + h][1]};
            const u32x2 a2 = {v[2 * h], v[2 * h + 1]};
            const long bl8 = __builtin_bit_cast(long, b2), al8 = __builtin_bit_cast(long, a2);
            constexpr bool WB = P::KIND == DK_E5M2, AB = (P::FLAGS & FL_ABF8) != 0;
            if constexpr (!WB && !AB) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bl8, al8, acc[mf], 0, 0, 0);
            if constexpr (!WB && AB) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(bl8, al8, acc[mf], 0, 0, 0);
            if constexpr (WB && !AB) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(bl8, al8, acc[mf], 0, 0, 0);
            if constexpr (WB && AB) acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(bl8, al8, acc[mf], 0, 0, 0);
          }
        } else {
          const u32x4 bv = {bfrag[g][0], bfrag[g][1], bfrag[g][2], bfrag[g][3]};
          acc[mf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bv), __builtin_bit_cast(i32x4, v), acc[mf], 0, 0, 0);
        }
      }
    }
  };

  // wave w takes a contiguous run of k-steps (a multiple of 4 long, so its blocks of 4 steps line up with
  // the 8-byte metadata loads); loads are unconditional (clamped step), compute is guarded
  const int nsteps = a.nsteps;
  const int last = nsteps - 1;
  const int run = (((nsteps + NW - 1) / NW) + 3) & ~3;
  const int t_lo = wave * run;
  const int my_steps = t_lo >= nsteps ? 0 : (nsteps - t_lo < run ? nsteps - t_lo : run);   // wave-uniform
  auto meta_load = [&](int block_first_step, u32x2& gs, u32x2& gz) {
    int base = block_first_step < a.kg - 4 ? block_first_step : a.kg - 4;   // group index = step when wide
    base = base < 0 ? 0 : base;
    gs = *reinterpret_cast<const u32x2*>(Sp + srow + base);
    if constexpr (MODE == MD_ZO || MODE == MD_ZR) gz = *reinterpret_cast<const u32x2*>(Zp + srow + base);
    else gz = u32x2{0u, 0u};
  };
  Step ring[PF];
  static_assert(PF == 4, "blocks of four steps");
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int t = t_lo + i;
    load_step(t < nsteps ? t : last, ring[i]);
  }
  u32x2 gs_cur = {0u, 0u}, gz_cur = {0u, 0u}, gs_nxt = {0u, 0u}, gz_nxt = {0u, 0u};
  if (wide) meta_load(t_lo < nsteps ? t_lo : 0, gs_cur, gz_cur);
  for (int s0 = 0; s0 < my_steps; s0 += PF) {
    const bool more = s0 + PF < my_steps;                    // wave-uniform: short K issues nothing more
    if (wide && more) meta_load(t_lo + s0 + PF, gs_nxt, gz_nxt);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int s = s0 + i;
      if (wide) {                                            // step i of the block: half i of the 8 bytes
        ring[i].b.s[0] = (gs_cur[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        ring[i].b.z[0] = (gz_cur[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
      }
      if (s < my_steps) compute(ring[i]);
      const int tw = t_lo + s + PF;                          // refill the slot just consumed
      if (more) load_step(tw < nsteps ? tw : last, ring[i]);
    }
    gs_cur = gs_nxt;
    gz_cur = gz_nxt;
  }

  // ---- meet in LDS: slot [wave][mf][lane], summed in wave order by the threads of wave mf ----
  acc_t* red = reinterpret_cast<acc_t*>(smem_raw);
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) red[(wave * MF + mf) * 64 + lane] = acc[mf];
  __syncthreads();
  if (wave >= MF) return;
  acc_t sum = red[wave * 64 + lane];                        // wave 0's share of fragment `wave`
#pragma unroll
  for (int w = 1; w < NW; ++w) sum += red[(w * MF + wave) * 64 + lane];
  const int nb = n0 + kb * 4;
  const int m = wave * 16 + fr;
  if (nb < a.N && m < a.M) store_quad<P>(a, sum, m, nb);
}

// ------------------------------------------------------------------------------------------
// decode-batch member, activations through LDS (M <= 16, 2-byte or 1-byte activations).  Same decomposition as
// wq_gemm_decode_kernel - a workgroup per 16-row weight fragment owning all of K, its 8 waves taking contiguous
// k ranges, one launch, deterministic - but the activation operand no longer comes as fragment-shaped global loads
// (16 rows x 64 B per wave instruction: 16 cache-line halves, twice the texture-path work per byte, PMC:
// profiles/r02_pmc_before.json m16, TA busy 3.5x the GEMV's).  Each wave copies exactly the activation columns IT
// multiplies - 16 rows x its k range - into its own LDS region with global_load_lds (LDS-DMA: 1 KiB of contiguous
// row segments per instruction, no VGPRs, no ds_write), XOR-swizzled through the SOURCE address so that the MFMA
// operand reads (ds_read_b128) are conflict free; nothing is shared between waves, so the only synchronisation is
// the issuing wave's own vmcnt.  Blocks of 4 k-steps: the block's activations (only the row groups below M), then its
// metadata and weights, all in flight before any is used.
// Same-call A/B against the direct-load member, uint4 g128 + zeros, 4096^2 (profiles/r02_ab_decode_lds.txt):
// M=16 8.43 -> 6.66 us, M=12 7.8 -> 6.2, M=8 7.1 -> 5.95, M=3 6.2 -> 5.76; 3584x8192 M=16 14.4 -> 10.5; int2 x int8 M=16 5.68 -> 5.08;
// with non-temporal weight loads on top M=16 6.45, M=12 6.08, int2 x int8 4.87.
// What bounds it (tools/decode_trace.hip, profiles/r02_decode_trace.txt): every workgroup reads ALL of A - 256 x 128 KiB
// = 32 MiB through the L2s at M = 16 - and the LDS-DMA issue stalls for ~2.5 us on that; weights-first issue order
// and hand-kept per-k-step vmcnt waits were tried and measured no better (the wave is through its DMA queue only
// when everything else has long arrived).
// ------------------------------------------------------------------------------------------
template <class P, int KSL = 0>      // KSL = 1: the K-sliced form alone (its own instantiation: member 212) - every other form is compiled out of it;
                                     // KSL = 2: the same walk with ONE slice - the whole of K - per workgroup (member 213): a wave owns whole
                                     // fragments, adds all of K in one accumulator and stores the output itself: no meeting, no partial sums
__global__ void __launch_bounds__(P::THREADS) wq_gemm_decode_lds_kernel(const GemmArgs a) {
  using T = typename P::T;
  constexpr int NJ = P::NJ, WL = P::WL, MODE = P::MODE, NW = P::NWAVES;
  constexpr bool F16 = P::AT == AT_F16, F8 = P::AT == AT_F8, FACC = F16 || F8;
  constexpr int ASZ = F16 ? 2 : 1;
  using acc_t = typename std::conditional<FACC, f32x4, i32x4>::type;
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;
  constexpr int ZPB = 8 / ZB;
  constexpr int PF = 4;                       // k-steps per block
  static_assert(P::NFW == 1 && P::MF == 1 && P::AT != AT_I4, "decode-LDS member: one weight fragment, M <= 16, unpacked activations");
  constexpr int STEP_BYTES = 16 * P::ROW_BYTES;          // one k-step of the 16-row activation fragment in LDS: 4 KiB
  constexpr int REGION = PF * STEP_BYTES;                // per wave: 16 KiB

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // every kernel argument in ONE scalar round trip (the compiler otherwise fetches them where first used: dependent
  // s_load waits in front of the first load of a kernel whose whole life is ~4 us; see wqaa_gemvx_kernel.h)
  asm volatile("" ::"s"(a.A), "s"(a.B), "s"(a.scale), "s"(a.zeros), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.kg), "s"(a.gq_shift),
               "s"(a.row_bytes), "s"(a.nsteps), "s"(a.is_signed), "s"((int)gridDim.x));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  WQ_TRACE_DECL;
  WQ_TRACE(0);
  const int fr = lane & 15, kb = lane >> 4;
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // PERSISTENT form (round 4): a grid smaller than the number of 16-row weight fragments - the launcher caps it at one
  // workgroup per CU when N / 16 exceeds the chip AND every wave's k-range is one block (K <= NW * PF k-steps) - makes a
  // workgroup take fragments blk, blk + grid, ...: the activations are staged ONCE (a wave's whole k-range sits in its LDS
  // region), the next fragment's weights are in flight while this one is multiplied and reduced.  11008 x 4096 at M = 3 ... 16
  // was 2.7 rounds of one-fragment workgroups (or the split-K skinny member + its reduce launch: 13-15 us).
  const int nfrags = (a.N + 15) >> 4;
  const bool persistent = (int)gridDim.x < nfrags;
  int n0 = blk * 16;
  int nrow = n0 + fr;
  nrow = nrow < a.N ? nrow : a.N - 1;

  const uint8_t* Ap = reinterpret_cast<const uint8_t*>(a.A);
  const uint8_t* Bp = reinterpret_cast<const uint8_t*>(a.B);
  const uint16_t* Sp = reinterpret_cast<const uint16_t*>(a.scale);
  const uint16_t* Zp = reinterpret_cast<const uint16_t*>(a.zeros);
  const uint8_t* Qp = reinterpret_cast<const uint8_t*>(a.zeros);
  const uint8_t* brow = Bp + (long)nrow * a.row_bytes + (long)kb * (WL * 4);
  long srow = (long)nrow * a.kg;
  auto set_fragment = [&](int frag) __attribute__((always_inline)) {                           // the weight fragment the loads / the store that follow refer to
    n0 = frag * 16;
    nrow = n0 + fr;
    nrow = nrow < a.N ? nrow : a.N - 1;
    brow = Bp + (long)nrow * a.row_bytes + (long)kb * (WL * 4);
    srow = (long)nrow * a.kg;
  };
  unsigned char* region = smem_raw + wave * REGION;

  // LDS-DMA source of this lane for instruction q (rows 4q .. 4q+3) of a k-step: the slot it fills is (row, p = lane & 15);
  // slot p of row r holds the granule whose (j, kb) index is p ^ (r & 15)  [granule order (j << 2) | kb, see wq_gemm_kernel]
  uint32_t dsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int r = 4 * q + (lane >> 4);
    const int x = (lane & 15) ^ (r & 15);
    const int ns = (x & 3) * 4 + (x >> 2);                     // natural granule (kb' * 4 + j') inside the k-step
    r = r < a.M ? r : a.M - 1;                                  // rows >= M are never stored
    dsrc[q] = (uint32_t)r * (uint32_t)(a.K * ASZ) + (uint32_t)(ns * 16);
  }
  constexpr int ASTEP = P::KS * ASZ;                            // bytes of one k-step in a row of A (= 256)
  const int nq = (a.M + 3) >> 2;                                // row groups that hold real rows (the rest of the fragment is never stored)
  auto dma_step = [&](int t, int s) __attribute__((always_inline)) {                           // k-step t -> block slot s
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nq)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ap + (long)t * ASTEP + dsrc[q]),
                                         (__attribute__((address_space(3))) void*)(region + s * STEP_BYTES + q * 1024), 16, 0, 0);
  };
  // the same as instructions the compiler does NOT see (M0 = the LDS address, saved and put back).  With the builtin its scoreboard
  // makes the first LDS read wait for EVERY load in flight - LDS-DMA and register loads do not retire in one order as far as it
  // knows - so nothing could be asked for ahead of a tile.  Loads do return in order: whoever issues these counts the waits
  // (an explicit s_waitcnt that leaves only loads YOUNGER than the tile outstanding) and keeps the LDS reads below that wait.
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  auto dma_step_unseen = [&](int t, int slot_off) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nq) {
        const unsigned char* src = Ap + (long)t * ASTEP + dsrc[q];
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(wave * REGION + slot_off + q * 1024));
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
  };

  constexpr bool WIDE_OK = MODE == MD_S || MODE == MD_ZO || MODE == MD_ZR;
  const bool wide = WIDE_OK && a.gq_shift == 2 && (a.kg & 3) == 0;   // one group per k-step: 8-byte metadata loads per block
  auto w_load = [&](int t, BLane<P>& b) __attribute__((always_inline)) {
    const int kidx = t * 4 + kb;
    int gi = 0;
    if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (kidx >> a.gq_shift) : (int)__umulhi((uint32_t)kidx, a.gq_magic);
    load_lane_words<WL, true>(brow + (long)t * (4 * WL * 4), b.w[0]);
    if (!wide) {
      if constexpr (MODE != MD_NONE) b.s[0] = Sp[srow + gi];
      if constexpr (MODE == MD_ZO || MODE == MD_ZR) b.z[0] = Zp[srow + gi];
    }
    if constexpr (MODE == MD_ZQ) b.z[0] = Qp[(long)gi * a.zq_row_bytes + nrow / ZPB];
  };

  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
  cx.flip = 0u;
  if (P::KIND == DK_INT1 && a.is_signed) cx.flip = 0xFFFFFFFFu;
  if (P::KIND == DK_INT8 && a.is_signed) cx.flip = 0x80808080u;
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  const uint32_t zp4 = (!F16 && a.is_signed && T::SUBBYTE) ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) lut = make_fp4_lut(P::BF);
    else lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
  }

  acc_t acc = acc_t{0, 0, 0, 0};
  int zq_row = nrow;                                            // (packed zero points: the row the fragment in hand was loaded for)
  auto compute = [&](const BLane<P>& bl, int slot_off) __attribute__((always_inline)) {     // slot_off: the k-step's slot in the wave's region, bytes
    uint32_t bfrag[NJ][4];
    if constexpr (F16) {
      half_t zf = cx.zf;
      if constexpr (MODE == MD_ZQ) {
        const uint32_t zq = (bl.z[0] >> ((zq_row % ZPB) * ZB)) & ((1u << ZB) - 1u);
        zf = (half_t)(float)zq;
      }
      const half2_t s2 = MODE != MD_NONE ? splat(bits_to_half(bl.s[0])) : splat((half_t)1.0f);
      const half2_t z2 = (MODE == MD_ZO || MODE == MD_ZR) ? splat(bits_to_half(bl.z[0])) : splat((half_t)0.0f);
      if constexpr (P::BF)
        dequant_lane_bf16<P>(bl.w[0], (float)zf, MODE != MD_NONE ? bf16_bits_to_float(bl.s[0]) : 1.f, a.is_signed != 0, cx.flip, lut, bfrag,
                             (MODE == MD_ZO || MODE == MD_ZR) ? bf16_bits_to_float(bl.z[0]) : 0.f);
      else
        dequant_lane_f16<P>(bl.w[0], zf, s2, z2, cx, lut, bfrag);
    } else if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bfrag[j][0] = bl.w[0][2 * j];
        bfrag[j][1] = bl.w[0][2 * j + 1];
      }
    } else {
      dequant_lane_i8<P>(bl.w[0], zp4, cx.flip, bfrag);
    }
    // The LDS-DMA writes of this block were issued BEFORE its weight loads and loads return in order, so once the
    // weight word is here the activations are too.  The compiler does not know the ds_reads depend on the DMA: tie
    // their address to the weight word (an empty asm that "rewrites" the offset after reading the word), so the
    // reads are ordered after the vmcnt wait it inserts for the weights.
    uint32_t roff = (uint32_t)(slot_off + fr * P::ROW_BYTES);
    asm volatile("" : "+v"(roff) : "v"(bl.w[0][0]));
    const unsigned char* rowp = region + roff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(rowp + ((((g << 2) | kb) ^ fr) * 16));
      if constexpr (F16) {
        const u32x4 bv = {bfrag[g][0], bfrag[g][1], bfrag[g][2], bfrag[g][3]};
        if constexpr (P::BF)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, v), acc, 0, 0, 0);
        else
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv), __builtin_bit_cast(half8_t, v), acc, 0, 0, 0);
      } else if constexpr (F8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x2 b2 = {bfrag[2 * g + h][0], bfrag[2 * g + h][1]};
          const u32x2 a2 = {v[2 * h], v[2 * h + 1]};
          const long bl8 = __builtin_bit_cast(long, b2), al8 = __builtin_bit_cast(long, a2);
          constexpr bool WB = P::KIND == DK_E5M2, AB = (P::FLAGS & FL_ABF8) != 0;
          if constexpr (!WB && !AB) acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bl8, al8, acc, 0, 0, 0);
          if constexpr (!WB && AB) acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(bl8, al8, acc, 0, 0, 0);
          if constexpr (WB && !AB) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(bl8, al8, acc, 0, 0, 0);
          if constexpr (WB && AB) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(bl8, al8, acc, 0, 0, 0);
        }
      } else {
        const u32x4 bv = {bfrag[g][0], bfrag[g][1], bfrag[g][2], bfrag[g][3]};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bv), __builtin_bit_cast(i32x4, v), acc, 0, 0, 0);
      }
    }
  };

  // wave w takes a contiguous run of k-steps, a multiple of 4 long (blocks line up with the 8-byte metadata loads)
  const int nsteps = a.nsteps;
  const int last = nsteps - 1;
  const int run = KSL == 2 ? ((nsteps + 3) & ~3) : ((((nsteps + NW - 1) / NW) + 3) & ~3);
  // (K-sliced form: the eighth of K is the WORKGROUP's - block % 8, one slice per XCD - and every wave of it walks that range)
  // K-sliced form, workgroup -> (slice, group): a band of ROWS per XCD - the eight slices of a group run on the group's XCD (block = xcd +
  // 8 (slice + 8 (group / 8)), group = xcd + 8 (group / 8)) - where the groups come in eights; else slice = block % 8.  (One SLICE per
  // XCD measured 3-10 % slower - at K = 8192, where an XCD would read 512 B of every 4 KiB row, two of its sixteen L2 channels:
  // profiles/r05_ab_kslice.txt)
  int ksl_slice = (int)blockIdx.x & 7, ksl_grp = (int)blockIdx.x >> 3;
  if (KSL == 2) {
    ksl_slice = 0;
    ksl_grp = blk;                                  // (the XCD swizzle above: a band of row groups per XCD)
  } else if (KSL == 1 && (gridDim.x & 63) == 0) {
    const int j = (int)blockIdx.x >> 3;
    ksl_slice = j & 7;
    ksl_grp = ((int)blockIdx.x & 7) + 8 * (j >> 3);
  }
  const int t_lo = (KSL != 0 ? ksl_slice : wave) * run;
  const int my_steps = t_lo >= nsteps ? 0 : (nsteps - t_lo < run ? nsteps - t_lo : run);   // wave-uniform
  acc_t* red = reinterpret_cast<acc_t*>(smem_raw + NW * REGION);
  // ---- the hand-counted forms: 4-bit weights with one Scale / Zeros group per k-step (the headline formats) --------------------------
  // PERSISTENT (the launcher guarantees run <= PF - one block per wave - and at most six fragments per workgroup): the walk in
  // straight line with every load an inline-assembly instruction and every wait counted by hand.  With compiler-tracked loads the
  // first read of the staged tile waited for ALL loads in flight - LDS-DMA and register loads do not retire in one order, so the
  // compiler assumes the worst - and a refill behind a test ended in a copy that did the same: 11.6 us at 11008 x 4096 where the
  // weight stream needs ~6.  Here the tile and ALL the workgroup's fragments are asked for at once and fragment i is multiplied
  // while i + 1, i + 2 are still arriving.  A wave's own vmcnt retires in order; wave 0's stores in between only make a wait stricter.
  // WHOLE TILE (a.decode_long, K > 4096): the wave's k-range is NBK = 2 or 3 blocks; its activations fit the region when a k-step's
  // slot holds only the row groups below M (nq KiB instead of 4: the MFMA's rows >= 4 nq read the slots behind - never stored), so
  // the tile is staged once here too and the walk is over UNITS (fragment, block), three in flight, fragment-major - instead of
  // blocks drained one by one at ~2 TB/s (profiles/r04_decode_longk.txt).  Same k order per wave, same meeting: bit-identical.
  if constexpr (WL == 4 && WIDE_OK) {
    // (8-byte metadata loads as instructions need 4-byte alignment only: K / g even.  Where K / g is not a multiple of 4 the last
    // block's load is moved back to end inside the row and the halves are taken `sh` groups further on)
    const bool wide_c = a.gq_shift == 2 && (a.kg & 1) == 0;
    if (KSL != 0 || (wide_c && (persistent || a.decode_long))) {      // (KSL: the launcher checked the metadata's alignment)
      constexpr bool ZP = MODE == MD_ZO || MODE == MD_ZR;
      constexpr int NOPS = PF + 1 + (ZP ? 1 : 0);        // loads per unit and lane
      struct AF {
        u32x4 w[PF];
        u32x2 gs, gz;
        int row, sh;
      };
      const int G = (int)gridDim.x;
      const int n_own = (nfrags - 1 - blk) / G + 1;      // fragments of this workgroup
      auto issue_blk = [&](int frag, int j, AF& f) __attribute__((always_inline)) {          // block j of the wave's k-range
        set_fragment(frag);
        f.row = nrow;
        const int t0 = t_lo + j * PF;
        int base = t0 < a.kg - 4 ? t0 : a.kg - 4;
        base = base < 0 ? 0 : base;
        f.sh = (t0 - base) & 3;
        const uint16_t* sp = Sp + srow + base;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(f.gs) : "v"(sp) : "memory");
        if constexpr (ZP) {
          const uint16_t* zp = Zp + srow + base;
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(f.gz) : "v"(zp) : "memory");
        } else {
          f.gz = u32x2{0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          int t = t0 + i;
          t = t < nsteps ? t : last;
          const uint8_t* wp = brow + (long)t * (4 * WL * 4);
          asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(f.w[i]) : "v"(wp) : "memory");
        }
      };
      auto issue = [&](int frag, AF& f) __attribute__((always_inline)) { issue_blk(frag, 0, f); };
      // the wait hands the fragment's registers on: nothing that reads them can be scheduled above it.
      // CONTRACT with the compiler: between a load and its wait the destination registers must stay where they are - a spill
      // or an out-of-line call (captures on the stack) would copy them before the data is there.  Hence every lambda of this
      // kernel is always_inline, and tests/test_abi.py::test_counted_decode_members_keep_their_loads_in_registers reads the
      // built library's metadata: no scratch, no stack in any instantiation that takes this path; tools/check_vmem_hazards.py
      // (tests/test_vmem_hazard_checker.py) walks their disassembly: nothing touches a register a load is still to write.  (The same walk with
      // compiler-tracked loads needs no contract and was measured ~0.8 us slower at 11008 x 4096: where the one-, two- and
      // three-fragment paths share their first loads the compiler's count falls back to vmcnt(0).)
      auto landed = [&](auto NY, AF& f) __attribute__((always_inline)) {
        constexpr int ny = decltype(NY)::value;          // loads issued after this fragment's
        static_assert(PF == 4, "the operand list below");
        if constexpr (ZP)
          asm volatile("s_waitcnt vmcnt(%6)" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.gs), "+v"(f.gz) : "n"(ny) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%5)" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.gs) : "n"(ny) : "memory");
      };
      auto multiply_blk = [&](const AF& f, int j, int slot_bytes) __attribute__((always_inline)) {   // acc += block j (slots of slot_bytes)
        zq_row = f.row;
        const uint64_t s64 = (((uint64_t)f.gs[1] << 32) | f.gs[0]) >> (16 * f.sh);
        const uint64_t z64 = (((uint64_t)f.gz[1] << 32) | f.gz[0]) >> (16 * f.sh);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          BLane<P> bl;
          bl.w[0][0] = f.w[i][0]; bl.w[0][1] = f.w[i][1]; bl.w[0][2] = f.w[i][2]; bl.w[0][3] = f.w[i][3];
          bl.s[0] = (uint32_t)(s64 >> (16 * i)) & 0xFFFFu;
          bl.z[0] = (uint32_t)(z64 >> (16 * i)) & 0xFFFFu;
          if (j * PF + i < my_steps) compute(bl, (j * PF + i) * slot_bytes);
        }
      };
      auto multiply = [&](const AF& f) __attribute__((always_inline)) -> acc_t {
        acc = acc_t{0, 0, 0, 0};
        multiply_blk(f, 0, STEP_BYTES);
        return acc;
      };
      // the waves meet once per batch of (up to) three fragments, with their partial sums of the whole batch (three sets of
      // slots), and waves 0, 1, 2 sum and store one fragment each - in wave order, as the one-fragment form does
      auto meet = [&](const acc_t& p0, const acc_t& p1, const acc_t& p2, int first, int count) __attribute__((always_inline)) {
        red[wave * 64 + lane] = p0;
        red[(NW + wave) * 64 + lane] = p1;
        red[(2 * NW + wave) * 64 + lane] = p2;
        __syncthreads();
        if (wave < count) {
          const acc_t* r = red + wave * (NW * 64);
          acc_t sum = r[lane];
#pragma unroll
          for (int w = 1; w < NW; ++w) sum += r[w * 64 + lane];
          const int nb = (first + wave * G) * 16 + kb * 4;
          if (nb < a.N && fr < a.M) store_quad<P>(a, sum, fr, nb);
        }
      };
      AF f0, f1, f2;
      const acc_t zero = acc_t{0, 0, 0, 0};
      acc_t p0 = zero, p1 = zero, p2 = zero;
      if constexpr (KSL != 0) {
        static_assert(F16 && !P::BF, "the K-sliced form: float16 activations");
        {
          // K-SLICED (round 5, long K): the forms above make every workgroup read ALL of A - at K = 28672 twice its weights' bytes, and a CU
          // ingests ~40 GB/s: 8192 x 28672 ran at 2.5 TB/s.  Here the k-range that was a WAVE's is a WORKGROUP's: workgroup
          // (slice, group) stages rows < M of slice
          // `slice` once, in M-sized slots SHARED by its 8 waves (run x nq KiB), and every wave walks its own weight fragments
          // (group * 8 + wave, + waves-per-slice, ...) through the slice's blocks - units (fragment, block), three in flight, as the
          // whole-tile form - keeping ONE accumulator per fragment in k order: exactly the partial sum wave `slice` of the one-launch form
          // holds.  It leaves as 1 KiB of fp32 (ws[fragment][slice][lane]) and wq_mid_reduce_kernel adds slices 0 .. 7 in that order:
          // bit-identical to `xdl`.  Waits are counted: a unit's wait leaves the loads of the two units asked for after it outstanding;
          // beyond the wave's last unit the asks go on (fragment index past the end = row N - 1 for all 16 rows: one cache line per
          // instruction) so that the count holds to the end without a branch around a load.
          const int slice = ksl_slice, grp = ksl_grp;
          const int slot_bytes = nq * 1024;
          const int nbk = run / PF;
          region = smem_raw;
          for (int sidx = wave; sidx < run; sidx += NW) {
            const int t = t_lo + sidx;
            const unsigned char* srcb = Ap + (long)(t < nsteps ? t : last) * ASTEP;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < nq) {
                const unsigned char* src = srcb + dsrc[q];
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(sidx * slot_bytes + q * 1024));
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
              }
          }
          const int wps = (KSL == 2 ? (int)gridDim.x : ((int)gridDim.x >> 3)) * NW;            // waves per slice
          // (whole-K form: fragment = wave x groups + group - with fewer fragments than waves every CU still gets its share of them)
          const int first = KSL == 2 ? wave * (int)gridDim.x + grp : grp * NW + wave;
          const int nfr = first < nfrags ? (nfrags - 1 - first) / wps + 1 : 0;
          const int U = nfr * nbk;
          int qf = first, qj = 0;                                // the unit asked for next
          auto ask = [&](AF& f) __attribute__((always_inline)) {
            issue_blk(qf, qj, f);
            if (++qj == nbk) {
              qj = 0;
              qf += wps;
            }
          };
          ask(f0);
          ask(f1);
          ask(f2);
          WQ_TRACE(1);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NOPS) : "memory");     // this wave's share of the slice is in LDS ...
          __builtin_amdgcn_sched_barrier(0);
          WQ_TRACE(2);
          __builtin_amdgcn_s_barrier();                                        // ... and everybody else's
          __builtin_amdgcn_sched_barrier(0);
          WQ_TRACE(3);
          f32x4* wsq = reinterpret_cast<f32x4*>(a.ws);
          int cf = first, cj = 0, done = 0;                      // the unit multiplied next
          auto unit = [&](AF& f) __attribute__((always_inline)) {
            WQ_TRACE_WAIT_BEGIN;
            landed(std::integral_constant<int, 2 * NOPS>{}, f);
            WQ_TRACE_WAIT_END(6);
            if (done < U) {
              if (cj == 0) acc = zero;
              multiply_blk(f, cj, slot_bytes);
              if (++cj == nbk) {
                if constexpr (KSL == 2) {
                  const int nb = cf * 16 + kb * 4;                 // the whole sum: this wave stores the fragment's outputs
                  if (nb < a.N && fr < a.M) store_quad<P>(a, acc, fr, nb);
                } else {
                  wsq[((long)cf * 8 + slice) * 64 + lane] = acc;
                }
                cj = 0;
                cf += wps;
              }
            }
            ++done;
            WQ_TRACE_IF(done == 1, 4);
            ask(f);
          };
          for (int u = 0; u < U; u += 3) {
            unit(f0);
            unit(f1);
            unit(f2);
          }
          WQ_TRACE(5);
          WQ_TRACE_DRAIN;          // (trace builds: the asks beyond the last unit are still in flight and the dump re-uses their registers)
          WQ_TRACE_DUMP(NW);
          return;
        }
      }
      if (a.decode_long) {
        const int slot_bytes = nq * 1024;
        // units u = (fragment u / NBK, block u % NBK), registers u % 3; a wait counts the loads of the units issued after u
        auto long_walk = [&](auto NOWN_, auto NBK_) __attribute__((always_inline)) {
          constexpr int NOWN = decltype(NOWN_)::value, NBK = decltype(NBK_)::value, U = NOWN * NBK;
          static_assert(U <= 6 && NOWN <= 3, "units");
#pragma unroll
          for (int sidx = 0; sidx < NBK * PF; ++sidx) {
            const int t = t_lo + sidx;
            dma_step_unseen(t < nsteps ? t : last, sidx * slot_bytes);
          }
          auto regs = [&](auto UC) __attribute__((always_inline)) -> AF& {
            constexpr int r = decltype(UC)::value % 3;
            if constexpr (r == 0) return f0;
            else if constexpr (r == 1) return f1;
            else return f2;
          };
          auto ask = [&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value;
            if constexpr (u < U) issue_blk(blk + (u / NBK) * G, u % NBK, regs(UC));
          };
          auto unit = [&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value;
            if constexpr (u < U) {
              constexpr int after = (U - 1 < u + 2 ? U - 1 : u + 2) - u;
              landed(std::integral_constant<int, after * NOPS>{}, regs(UC));
              if constexpr (u % NBK == 0) acc = zero;
              multiply_blk(regs(UC), u % NBK, slot_bytes);
              if constexpr (u % NBK == NBK - 1) {
                if constexpr (u / NBK == 0) p0 = acc;
                else if constexpr (u / NBK == 1) p1 = acc;
                else p2 = acc;
              }
              ask(std::integral_constant<int, u + 3>{});
            }
          };
          ask(std::integral_constant<int, 0>{});
          ask(std::integral_constant<int, 1>{});
          ask(std::integral_constant<int, 2>{});
          unit(std::integral_constant<int, 0>{});
          unit(std::integral_constant<int, 1>{});
          unit(std::integral_constant<int, 2>{});
          unit(std::integral_constant<int, 3>{});
          unit(std::integral_constant<int, 4>{});
          unit(std::integral_constant<int, 5>{});
          meet(p0, p1, p2, blk, NOWN);
        };
        using std::integral_constant;
        const int nbk = run / PF;                            // (the launcher: nbk = 2, 3; nbk * PF * nq <= 16 slots of 1 KiB; n_own * nbk <= 6)
        if (nbk == 2) {
          if (n_own == 1) long_walk(integral_constant<int, 1>{}, integral_constant<int, 2>{});
          else if (n_own == 2) long_walk(integral_constant<int, 2>{}, integral_constant<int, 2>{});
          else long_walk(integral_constant<int, 3>{}, integral_constant<int, 2>{});
        } else {                                             // nbk == 3
          if (n_own == 1) long_walk(integral_constant<int, 1>{}, integral_constant<int, 3>{});
          else long_walk(integral_constant<int, 2>{}, integral_constant<int, 3>{});
        }
        return;
      }
      // (persistent) the tile: one block per wave
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int t = t_lo + i;
        dma_step_unseen(t < nsteps ? t : last, i * STEP_BYTES);
      }
      if (n_own > 3) {
        // four to six fragments: the second batch refills the first one's registers as they are consumed.  NB2 = fragments of the
        // second batch; a wait counts the loads issued after the fragment it is for
        auto two_batches = [&](auto NB2) __attribute__((always_inline)) {
          constexpr int nb2 = decltype(NB2)::value;
          issue(blk, f0);
          issue(blk + G, f1);
          issue(blk + 2 * G, f2);
          landed(std::integral_constant<int, 2 * NOPS>{}, f0);
          p0 = multiply(f0);
          issue(blk + 3 * G, f0);
          landed(std::integral_constant<int, 2 * NOPS>{}, f1);
          p1 = multiply(f1);
          if constexpr (nb2 >= 2) issue(blk + 4 * G, f1);
          landed(std::integral_constant<int, (1 + (nb2 >= 2 ? 1 : 0)) * NOPS>{}, f2);
          p2 = multiply(f2);
          if constexpr (nb2 >= 3) issue(blk + 5 * G, f2);
          meet(p0, p1, p2, blk, 3);
          acc_t q0 = zero, q1 = zero, q2 = zero;
          landed(std::integral_constant<int, (nb2 - 1) * NOPS>{}, f0);
          q0 = multiply(f0);
          if constexpr (nb2 >= 2) {
            landed(std::integral_constant<int, (nb2 - 2) * NOPS>{}, f1);
            q1 = multiply(f1);
          }
          if constexpr (nb2 >= 3) {
            landed(std::integral_constant<int, 0>{}, f2);
            q2 = multiply(f2);
          }
          __syncthreads();                               // the first batch's slots have been read
          meet(q0, q1, q2, blk + 3 * G, nb2);
        };
        if (n_own == 4) two_batches(std::integral_constant<int, 1>{});
        else if (n_own == 5) two_batches(std::integral_constant<int, 2>{});
        else two_batches(std::integral_constant<int, 3>{});
        return;
      }
      if (n_own >= 3) {
        issue(blk, f0);
        issue(blk + G, f1);
        issue(blk + 2 * G, f2);
        landed(std::integral_constant<int, 2 * NOPS>{}, f0);
        p0 = multiply(f0);
        landed(std::integral_constant<int, NOPS>{}, f1);
        p1 = multiply(f1);
        landed(std::integral_constant<int, 0>{}, f2);
        p2 = multiply(f2);
      } else if (n_own == 2) {
        issue(blk, f0);
        issue(blk + G, f1);
        landed(std::integral_constant<int, NOPS>{}, f0);
        p0 = multiply(f0);
        landed(std::integral_constant<int, 0>{}, f1);
        p1 = multiply(f1);
      } else {
        issue(blk, f0);
        landed(std::integral_constant<int, 0>{}, f0);
        p0 = multiply(f0);
      }
      meet(p0, p1, p2, blk, n_own);
      return;
    }
  }
  if (persistent) {
    struct FragLoad {
      BLane<P> bs[PF];
      u32x2 gs, gz;
      int row;
    };
    auto load_fragment = [&](int frag, FragLoad& f) {
      set_fragment(frag);
      f.row = nrow;
      f.gs = u32x2{0u, 0u};
      f.gz = u32x2{0u, 0u};
      if (wide) {
        int base = t_lo < a.kg - 4 ? t_lo : a.kg - 4;
        base = base < 0 ? 0 : base;
        f.gs = *reinterpret_cast<const u32x2*>(Sp + srow + base);
        if constexpr (MODE == MD_ZO || MODE == MD_ZR) f.gz = *reinterpret_cast<const u32x2*>(Zp + srow + base);
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int t = t_lo + i;
        w_load(t < nsteps ? t : last, f.bs[i]);
      }
    };
    auto finish_fragment = [&](int frag, FragLoad& f, int parity) {
      acc = acc_t{0, 0, 0, 0};
      zq_row = f.row;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        if (wide) {
          f.bs[i].s[0] = (f.gs[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
          f.bs[i].z[0] = (f.gz[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        }
        if (i < my_steps) compute(f.bs[i], i * STEP_BYTES);
      }
      // two sets of meeting slots: the waves may be a fragment ahead of the one that sums
      acc_t* r = red + parity * (NW * 64);
      r[wave * 64 + lane] = acc;
      __syncthreads();
      if (wave == 0) {
        acc_t sum = r[lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += r[w * 64 + lane];
        const int nb = frag * 16 + kb * 4;
        if (nb < a.N && fr < a.M) store_quad<P>(a, sum, fr, nb);
      }
    };
    // the activations, once
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int t = t_lo + i;
      dma_step(t < nsteps ? t : last, i);
    }
    asm volatile("" ::: "memory");
    // up to three fragments' weights asked for ahead (12.2 us with one ahead, 11.6 with three, 13.3 before: 11008 x 4096 M = 3;
    // profiles/r04_ab_decode_persistent.txt).  The loads of a refill sit behind a test, and the copy at the join waits for
    // everything in flight; the branch-free form (clamped fragment indices, every load unconditional) was built and measured
    // WORSE - 18 us: the redundant fetches of the clamped fragment and a 6-way unrolled trip - and is not kept.
    FragLoad f0, f1, f2;
    const int G = (int)gridDim.x;
    int g0 = blk, g1 = blk + G, g2 = blk + 2 * G;
    load_fragment(g0, f0);
    if (g1 < nfrags) load_fragment(g1, f1);
    if (g2 < nfrags) load_fragment(g2, f2);
    int parity = 0;
    for (;;) {
      finish_fragment(g0, f0, parity);
      parity ^= 1;
      if (g1 >= nfrags) break;
      g0 += 3 * G;
      if (g0 < nfrags) load_fragment(g0, f0);
      finish_fragment(g1, f1, parity);
      parity ^= 1;
      if (g2 >= nfrags) break;
      g1 += 3 * G;
      if (g1 < nfrags) load_fragment(g1, f1);
      finish_fragment(g2, f2, parity);
      parity ^= 1;
      if (g0 >= nfrags) break;
      g2 += 3 * G;
      if (g2 < nfrags) load_fragment(g2, f2);
    }
    return;
  }
  // (Round 4, measured and NOT kept - tools/r04_decode_longk_probe.py, profiles/r04_decode_longk.txt: asking for the NEXT block's weights
  // before this block is multiplied, with the tile's DMA out of the compiler's sight and a marker load behind it so that the waits
  // stay counted.  Bit-identical and slower, 10.98 -> 12.08 us at 4096 x 11008 M = 4: loads return in order, so the tile - which
  // cannot be asked for before the previous block's LDS reads are done - returns behind every weight load asked for earlier;
  // one block ahead hides one block's multiply (~0.3 us) of a ~2 us round trip and costs the marker.)
  for (int s0 = 0; s0 < my_steps; s0 += PF) {
    BLane<P> bs[PF];
    u32x2 gs = {0u, 0u}, gz = {0u, 0u};
    // activations first (L2 resident after the first workgroups, and loads return in order), then the weights
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int t = t_lo + s0 + i;
      dma_step(t < nsteps ? t : last, i);                      // clamped: loads unconditional, compute guarded
    }
    asm volatile("" ::: "memory");
    if (wide) {
      int base = t_lo + s0 < a.kg - 4 ? t_lo + s0 : a.kg - 4;
      base = base < 0 ? 0 : base;
      gs = *reinterpret_cast<const u32x2*>(Sp + srow + base);
      if constexpr (MODE == MD_ZO || MODE == MD_ZR) gz = *reinterpret_cast<const u32x2*>(Zp + srow + base);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int t = t_lo + s0 + i;
      w_load(t < nsteps ? t : last, bs[i]);
    }
    WQ_TRACE_IF(s0 == 0, 1);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (wide) {
        bs[i].s[0] = (gs[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        bs[i].z[0] = (gz[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
      }
      if (s0 + i < my_steps) compute(bs[i], i * STEP_BYTES);
      WQ_TRACE_IF(s0 == 0 && i == 0, 2);
      WQ_TRACE_IF(s0 == 0 && i == PF - 2, 3);
    }
    // the region is rewritten by the next block's DMA: every ds_read of this block must have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- meet in LDS: slot [wave][lane] (its own region, after the activation ones), summed in wave order by wave 0 ----
  WQ_TRACE(4);
  red[wave * 64 + lane] = acc;
  __syncthreads();
  WQ_TRACE(5);
  if (wave != 0) {
    WQ_TRACE_DUMP(NW);
    return;
  }
  acc_t sum = red[lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) sum += red[w * 64 + lane];
  const int nb = n0 + kb * 4;
  if (nb < a.N && fr < a.M) store_quad<P>(a, sum, fr, nb);
  WQ_TRACE(6);
  WQ_TRACE_DUMP(NW);
}

// ------------------------------------------------------------------------------------------
// split-K reduction: C[m][n..n+3] = cast(sum_s ws[s][m][n..n+3]) (+ bias after the cast)
// ------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) wq_splitk_reduce_kernel(const void* ws_, void* C, const void* bias, int M, int N,
                                                               int ksplit, int out_dtype, int has_bias,
                                                               const float* epi_row, float epi_tensor) {
  using acc_t = typename std::conditional<F16, f32x4, i32x4>::type;
  const long quads = (long)M * N / 4;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= quads) return;
  const acc_t* ws = reinterpret_cast<const acc_t*>(ws_);
  // slices are read 4 at a time with independent loads (a plain loop waits for every 16-byte load before issuing the
  // next one); a slice beyond ksplit is not read at all (split 2 and 4 moved twice and four times the bytes they had to
  // when every round read 8 clamped slices).  Summation order: ((s0 + s1) + s2) + ... as before - bit-identical results.
  acc_t sum = acc_t{0, 0, 0, 0};
  for (int s0 = 0; s0 < ksplit; s0 += 4) {
    acc_t part[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      part[i] = acc_t{0, 0, 0, 0};
      if (s0 + i < ksplit) part[i] = __builtin_nontemporal_load(ws + (long)(s0 + i) * quads + q);     // read once
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (s0 + i < ksplit) sum += part[i];
  }
  const long base = q * 4;
  const int n = (int)(base % N);
  if constexpr (F16) {
    if (out_dtype == WQAA_F16 && has_bias != 2) {
      // the common output: the quad's four float16 results as ONE 8-byte store (was four 2-byte stores per thread)
      half_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = (half_t)sum[i];
        if (has_bias) v[i] = v[i] + (half_t)(float)reinterpret_cast<const half_t*>(bias)[n + i];
      }
      const half2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
      *reinterpret_cast<u32x2*>(reinterpret_cast<half_t*>(C) + base) = u32x2{as_u32(lo), as_u32(hi)};
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (F16) {
      float b = 0.f;   // has_bias: 1 = float16 bias, 2 = bfloat16 bias
      if (has_bias == 2) b = bf16_bits_to_float(reinterpret_cast<const uint16_t*>(bias)[n + i]);
      else if (has_bias) b = (float)reinterpret_cast<const half_t*>(bias)[n + i];
      store_out(C, base + i, sum[i], out_dtype, has_bias != 0, b);
    } else {
      if (epi_row) {
        store_out_fused(C, base + i, sum[i], epi_row[base / N], epi_tensor, has_bias != 0, bias, n + i);
      } else {
        const int b = has_bias ? (int)reinterpret_cast<const int8_t*>(bias)[n + i] : 0;
        store_out(C, base + i, sum[i], out_dtype, has_bias != 0, b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// B_decode to memory: the TE graph's first stage on its own (tirscript/matmul_dequantize_impl.py:391-449) - every weight
// decoded and (zero, scale)-dequantised by the SAME routines the MFMA members use in their loop (dequant_lane_*), written
// row-major (N, K) in A_dtype.  A thread owns one lane's k-block: WL packed words in, KL elements out (16-byte loads, NJ
// 16-byte stores; a wave reads 1 KiB and writes 4 KiB of one row, contiguous).  Used by the two-pass member (wqaa_gemm.hip)
// and by wqaa_dequantize.
// ------------------------------------------------------------------------------------------
template <class P>
__global__ void __launch_bounds__(256) wq_dequant_kernel(const GemmArgs a, void* out) {
  using T = typename P::T;
  constexpr int NJ = P::NJ, WL = P::WL, MODE = P::MODE;
  constexpr bool F16 = P::AT == AT_F16;
  static_assert(P::AT == AT_F16 || P::AT == AT_I8, "B_decode exists in float16 / bfloat16 / int8");
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;
  constexpr int ZPB = 8 / ZB;
  const int nkb = a.K / P::KL;                           // k-blocks per row
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)a.N * nkb) return;                       // whole waves when nkb % 64 == 0 (the exchange below needs that)
  const int n = (int)(id / nkb), kidx = (int)(id - (long)n * nkb);
  const uint8_t* Bp = reinterpret_cast<const uint8_t*>(a.B);
  uint32_t w[WL];
  load_lane_words<WL, true>(Bp + (long)n * a.row_bytes + (long)kidx * (WL * 4), w);      // read once: non-temporal
  int gi = 0;
  if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (kidx >> a.gq_shift) : (int)__umulhi((uint32_t)kidx, a.gq_magic);
  uint32_t sbits = 0, zbits = 0;
  if constexpr (MODE != MD_NONE) sbits = reinterpret_cast<const uint16_t*>(a.scale)[(long)n * a.kg + gi];
  if constexpr (MODE == MD_ZO || MODE == MD_ZR) zbits = reinterpret_cast<const uint16_t*>(a.zeros)[(long)n * a.kg + gi];
  if constexpr (MODE == MD_ZQ) zbits = reinterpret_cast<const uint8_t*>(a.zeros)[(long)gi * a.zq_row_bytes + n / ZPB];
  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
  cx.flip = 0u;
  if (P::KIND == DK_INT1 && a.is_signed) cx.flip = 0xFFFFFFFFu;
  if (P::KIND == DK_INT8 && a.is_signed) cx.flip = 0x80808080u;
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) lut = make_fp4_lut(P::BF);
    else lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
  }
  uint32_t frag[NJ][4];
  if constexpr (F16) {
    half_t zf = cx.zf;
    if constexpr (MODE == MD_ZQ) zf = (half_t)(float)((zbits >> ((n % ZPB) * ZB)) & ((1u << ZB) - 1u));
    const half2_t s2 = MODE != MD_NONE ? splat(bits_to_half(sbits)) : splat((half_t)1.0f);
    const half2_t z2 = (MODE == MD_ZO || MODE == MD_ZR) ? splat(bits_to_half(zbits)) : splat((half_t)0.0f);
    if constexpr (P::BF)
      dequant_lane_bf16<P>(w, (float)zf, MODE != MD_NONE ? bf16_bits_to_float(sbits) : 1.f, a.is_signed != 0, cx.flip, lut, frag,
                           (MODE == MD_ZO || MODE == MD_ZR) ? bf16_bits_to_float(zbits) : 0.f);
    else
      dequant_lane_f16<P>(w, zf, s2, z2, cx, lut, frag);
  } else {
    const uint32_t zp4 = (a.is_signed && T::SUBBYTE) ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
    dequant_lane_i8<P>(w, zp4, cx.flip, frag);
  }
  // A lane holds NJ consecutive 16-byte pieces of the row; written as they are, one store instruction of a wave would
  // touch 64 pieces NJ * 16 bytes apart (a quarter of every 64-byte segment).  Through LDS the wave's NJ * 64 pieces are
  // handed round so that store j of lane l writes piece j * 64 + l of the wave's contiguous NJ KiB: full lines.
  // (row ends are wave-aligned: K / KL k-blocks per row, 64 lanes per wave, K a multiple of the 4 * KL k-step)
  __shared__ u32x4 xch[256 * NJ];
  const int lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
  const bool whole_wave = (nkb & 63) == 0;                 // a wave never straddles two rows
  if (whole_wave) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xch[(wave0 + lane) * NJ + j] = u32x4{frag[j][0], frag[j][1], frag[j][2], frag[j][3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t* wbase = reinterpret_cast<uint8_t*>(out) + ((long)n * a.K + (long)(kidx - lane) * P::KL) * (F16 ? 2 : 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) reinterpret_cast<u32x4*>(wbase)[j * 64 + lane] = xch[wave0 * NJ + j * 64 + lane];
  } else {
    u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((long)n * a.K + (long)kidx * P::KL) * (F16 ? 2 : 1));
#pragma unroll
    for (int j = 0; j < NJ; ++j) dst[j] = u32x4{frag[j][0], frag[j][1], frag[j][2], frag[j][3]};
  }
}

typedef void (*gemm_fn)(const GemmArgs);

// member tables, one translation unit each (parallel builds): wqaa_gemm_inst_*.hip
gemm_fn pick_gemm_f16_int4(int layout, int mode, int mf);
gemm_fn pick_gemm_f16_int21(int kind, int layout, int mode, int mf);
gemm_fn pick_gemm_f16_other(int kind, int mode, int flags, int mf);
gemm_fn pick_gemm_bf16(int kind, int mode, int mf);
gemm_fn pick_gemm_i8_f8(int kind, int layout, int at, int flags, int mf);
gemm_fn pick_gemm_pp(int kind, int layout, int at, int mode, int flags, int bm, int bn, int* lds_bytes);   // wqaa_gemm_pp_kernel.h members

// mf codes: 1, 2, 4, 8 (16*mf x 128, 4 waves), 16 (256 x 256, 8 waves), 101/102/104 (skinny members),
// 201 (decode-batch member: one launch, K split across the waves of a workgroup), 211 (the same with the activations through LDS-DMA)
template <int KIND, int LAYOUT, int AT, int MODE, int FLAGS>
static gemm_fn pick_mf(int mf) {
  switch (mf) {
    case 16: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 16, 8>>;
    case 8: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 8>>;
    case 4: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 4>>;
    // (the pipelined 16- and 32-row members - codes 1 and 2 - are not instantiated: up to 64 rows the selector always takes the skinny
    // forms 101 / 102 / 104, the decode members or the mid-M member; a kernel census of the GPU suite, tools/kernel_census.py, found
    // all 166 of them launched by nothing - round 6)
    case 101: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 4, 1, 4>>;
    case 102: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 2, 4, 1, 4>>;
    case 104: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 4, 4, 1, 4>>;
    // 128-row skinny member: 8 waves x one weight fragment each (BN = 128), 4 k-steps per workgroup, every load first
    // (round 5's prune: the direct-load decode member is instantiated only where it is the selector's choice - packed int4 activations,
    // which the LDS-DMA member does not take; elsewhere it was reachable through a round-4 A/B aid alone)
    case 201: if constexpr (AT == AT_I4) return wq_gemm_decode_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 8, 1>>; else return nullptr;
    case 211: if constexpr (AT != AT_I4) return wq_gemm_decode_lds_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 8, 1>>; else return nullptr;
    // 212: the K-sliced form of 211 (long K; 4-bit weights x float16, Scale (+ Zeros) per 128: the hand-counted formats)
    case 212:
      if constexpr (AT == AT_F16 && (FLAGS & FL_BF16) == 0 && (KIND == DK_INT4 || KIND == DK_LUT4) && (MODE == MD_S || MODE == MD_ZO || MODE == MD_ZR))
        return wq_gemm_decode_lds_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 8, 1>, 1>;
      else return nullptr;
    // 213: a wave per fragment, the whole of K (wide outputs at K <= 4096 ... where the activation tile fits LDS)
    case 213:
      if constexpr (AT == AT_F16 && (FLAGS & FL_BF16) == 0 && (KIND == DK_INT4 || KIND == DK_LUT4) && (MODE == MD_S || MODE == MD_ZO || MODE == MD_ZR))
        return wq_gemm_decode_lds_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 8, 1>, 2>;
      else return nullptr;
    // 900: not a GEMM - B_decode to memory (two-pass member, wqaa_dequantize); launched with (GemmArgs, void* out)
    case 900:
      // (a native operator's B_decode is B itself: no member)
      if constexpr ((AT == AT_F16 || AT == AT_I8) && KIND != DK_NATIVE) return reinterpret_cast<gemm_fn>(wq_dequant_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1>>);
      else return nullptr;
    case 404: if constexpr (KIND == DK_INT4 && AT == AT_F16 && FLAGS == 0 && (MODE == MD_ZO || MODE == MD_ZR)) return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 4, 4, 2, 0, true>>; else return nullptr;
  }
  return nullptr;
}
template <int KIND, int LAYOUT>
static gemm_fn pick_modes(int mode, int mf) {
  switch (mode) {
    case MD_NONE: return pick_mf<KIND, LAYOUT, AT_F16, MD_NONE, 0>(mf);
    case MD_S: return pick_mf<KIND, LAYOUT, AT_F16, MD_S, 0>(mf);
    case MD_ZO: return pick_mf<KIND, LAYOUT, AT_F16, MD_ZO, 0>(mf);
    case MD_ZR: return pick_mf<KIND, LAYOUT, AT_F16, MD_ZR, 0>(mf);
    case MD_ZQ: return pick_mf<KIND, LAYOUT, AT_F16, MD_ZQ, 0>(mf);
  }
  return nullptr;
}
template <int KIND, int FLAGS>
static gemm_fn pick_modes_fp(int mode, int mf) {
  switch (mode) {
    case MD_NONE: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_NONE, FLAGS>(mf);
    case MD_S: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_S, FLAGS>(mf);
  }
  return nullptr;
}

}  // namespace wqaa
