// wqaa_gemm.hip - W_q x A MFMA GEMM family for gfx950 (M >= 8: the matrix-core-bound case).
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
  int ksplit;         // > 1: workgroup (tile, s) covers k-steps [s*nsteps/ksplit, (s+1)*nsteps/ksplit) and
  void* ws;           //      writes fp32 / int32 partial sums to ws[s][M][N]; a second kernel reduces
};

// ------------------------------------------------------------------------------------------
// policy: one k-step is KS = 4 * KL deep; a lane owns KL consecutive k of one weight row
// ------------------------------------------------------------------------------------------
template <int KIND_, int LAYOUT_, int AT_, int MODE_, int FLAGS_, int MF_, int NWAVES_ = 4, int NFW_ = 2, int SK_ = 0>
struct GemmPolicy {
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
  static_assert(AG >= 1 && AG * 64 * NWAVES_ == 16 * MF_ * 16, "tile / workgroup mismatch");
  static constexpr int BM = 16 * MF, BN = 16 * NFW * NWAVES;
  static constexpr bool STRICT = (FLAGS_ & FL_STRICT) != 0;
  using T = KindTraits<KIND_, AT_>;
  static constexpr int BITS = T::BITS;
  static constexpr int EPW = T::EPW;
  // elements of k per lane per MFMA, MFMAs per k-step, k per lane per k-step
  static constexpr int KPM = AT_ == AT_I8 ? 16 : 8;
  // MFMAs per 16-byte activation granule: fp8 operands are 8 bytes, so a granule feeds two
  static constexpr int MPG = AT_ == AT_F8 ? 2 : 1;
  static constexpr int NJ = 4 * MPG;
  static constexpr int KL = KPM * NJ;                 // 32 (fp16) / 64 (int8, fp8)
  static constexpr int KS = 4 * KL;                   // 128 / 256: one LDS row is 256 bytes either way
  static constexpr int WL = KL * BITS / 32;           // 32-bit weight words per lane per k-step
  static constexpr int ROW_BYTES = 256;
  static constexpr int LDS_BYTES = (SK_ > 0 ? SK_ : 2) * BM * ROW_BYTES;
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
        const uint32_t x = w[wi] ^ cx.flip;
        const half2_t off = splat(cx.off8 + zf);
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
      for (int i = 0; i < NQ; ++i) t[i] = sub_bytes(t[i], zp4);
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

template <int NW_>
__device__ __forceinline__ void load_lane_words(const uint8_t* p, uint32_t (&w)[NW_]) {
  if constexpr (NW_ % 4 == 0) {
#pragma unroll
    for (int q = 0; q < NW_ / 4; ++q) {
      const u32x4 v = reinterpret_cast<const u32x4*>(p)[q];
      w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
    }
  } else if constexpr (NW_ == 2) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    w[0] = v[0]; w[1] = v[1];
  } else {
    static_assert(NW_ == 1, "unsupported lane word count");
    w[0] = *reinterpret_cast<const uint32_t*>(p);
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
  constexpr int ASZ = F16 ? 2 : 1;                        // bytes per activation element
  using acc_t = typename std::conditional<FACC, f32x4, i32x4>::type;
  constexpr int ZB = T::SUBBYTE ? T::BITS : 8;
  constexpr int ZPB = 8 / ZB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15;          // fragment row (weight n / activation m)
  const int kb = lane >> 4;          // k-block of the lane

  // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous band of M tiles so
  // its L2 keeps one activation band and streams the (small, packed) weights
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int ntiles = a.tiles_m * a.tiles_n;
  const int split = blk / ntiles;        // k-slice of this workgroup (0 when ksplit == 1)
  blk -= split * ntiles;
  const int tile_m = blk / a.tiles_n, tile_n = blk % a.tiles_n;
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
  const uint8_t* aptr[AG];
  int a_lds_off[AG];
#pragma unroll
  for (int it = 0; it < AG; ++it) {
    const int gid = it * P::THREADS + tid;
    const int r = gid >> 4, q = gid & 15;
    const int ns = ((q & 7) >> 1) * 4 + (q & 1) + ((q >> 3) << 1);   // natural granule = kb' * 4 + j'
    int m = m0 + r;
    m = m < a.M ? m : a.M - 1;
    aptr[it] = Ap + (long)m * a.K * ASZ + ns * 16;
    const int phys = (((ns & 3) << 2) | (ns >> 2)) ^ (r & 15);
    a_lds_off[it] = r * P::ROW_BYTES + phys * 16;
  }
  auto a_load = [&](int t) {
    const long koff = (long)t * (P::KS * ASZ);
#pragma unroll
    for (int it = 0; it < AG; ++it) areg[it] = *reinterpret_cast<const u32x4*>(aptr[it] + koff);
  };
  auto a_store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < AG; ++it)
      *reinterpret_cast<u32x4*>(smem_raw + buf * (P::BM * P::ROW_BYTES) + a_lds_off[it]) = areg[it];
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
  cx.off8 = (half_t)(a.is_signed ? 1152.0f : 1024.0f);
  make_magic(cx.magic);
  const uint32_t zp4 = (!F16 && a.is_signed && T::SUBBYTE) ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) {
      lut = make_fp4_lut();
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

  };

  if constexpr (P::SK > 0) {
    constexpr int S = P::SK;
    const int t0 = split * S;
    u32x4 areg_s[S][AG];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int t = t0 + q < a.nsteps ? t0 + q : a.nsteps - 1;
      const long koff = (long)t * (P::KS * ASZ);
#pragma unroll
      for (int it = 0; it < AG; ++it) areg_s[q][it] = *reinterpret_cast<const u32x4*>(aptr[it] + koff);
    }
    BLane<P> bs[S];
#pragma unroll
    for (int q = 0; q < S; ++q) b_load(t0 + q < a.nsteps ? t0 + q : a.nsteps - 1, bs[q]);
#pragma unroll
    for (int q = 0; q < S; ++q)
#pragma unroll
      for (int it = 0; it < AG; ++it)
        *reinterpret_cast<u32x4*>(smem_raw + q * (P::BM * P::ROW_BYTES) + a_lds_off[it]) = areg_s[q][it];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < S; ++q)
      if (t0 + q < a.nsteps) compute_step(bs[q], smem_raw + q * (P::BM * P::ROW_BYTES));
  } else {
    const int t_begin = (int)((long)split * a.nsteps / a.ksplit);
    const int nsteps = (int)((long)(split + 1) * a.nsteps / a.ksplit);   // end of this workgroup's k range
    BLane<P> bcur, bnext;
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

  // ---- epilogue: D[i][col]: weight row n = nbase + kb * 4 + i, activation row m = mbase + fr ----
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
        ws[(((long)split * a.M + m) * a.N + nb) >> 2] = acc[mf][nf];
      }
    }
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
        if constexpr (F16) bias_f[i] = (float)reinterpret_cast<const half_t*>(a.bias)[nb + i];
        else if constexpr (F8) bias_f[i] = 0.f;   // the reference defines no fp8 bias operand
        else bias_i[i] = (int)reinterpret_cast<const int8_t*>(a.bias)[nb + i];
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
          *reinterpret_cast<u32x2*>(reinterpret_cast<half_t*>(a.C) + base) = u32x2{as_u32(lo), as_u32(hi)};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) store_out(a.C, base + i, acc[mf][nf][i], a.out_dtype, a.has_bias != 0, bias_f[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_out(a.C, base + i, acc[mf][nf][i], a.out_dtype, a.has_bias != 0, bias_i[i]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// split-K reduction: C[m][n..n+3] = cast(sum_s ws[s][m][n..n+3]) (+ bias after the cast)
// ------------------------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) wq_splitk_reduce_kernel(const void* ws_, void* C, const void* bias, int M, int N,
                                                               int ksplit, int out_dtype, int has_bias) {
  using acc_t = typename std::conditional<F16, f32x4, i32x4>::type;
  const long quads = (long)M * N / 4;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= quads) return;
  const acc_t* ws = reinterpret_cast<const acc_t*>(ws_);
  // slices are read 8 at a time with independent loads (clamped index, masked add): a plain loop
  // waits for every 16-byte load before issuing the next one
  acc_t sum = acc_t{0, 0, 0, 0};
  for (int s0 = 0; s0 < ksplit; s0 += 8) {
    acc_t part[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sl = s0 + i < ksplit ? s0 + i : ksplit - 1;
      part[i] = ws[(long)sl * quads + q];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (s0 + i < ksplit) sum += part[i];
  }
  const long base = q * 4;
  const int n = (int)(base % N);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (F16) {
      const float b = has_bias ? (float)reinterpret_cast<const half_t*>(bias)[n + i] : 0.f;
      store_out(C, base + i, sum[i], out_dtype, has_bias != 0, b);
    } else {
      const int b = has_bias ? (int)reinterpret_cast<const int8_t*>(bias)[n + i] : 0;
      store_out(C, base + i, sum[i], out_dtype, has_bias != 0, b);
    }
  }
}

// library-owned scratch for the partial sums (one per device, grown on demand, never shrunk).
// Growing calls hipMalloc: do the first call of a new shape outside stream capture.
static void* g_ws[16] = {nullptr};
static size_t g_ws_bytes[16] = {0};
static std::mutex g_ws_mu;
static void* workspace(size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (g_ws_bytes[dev] < bytes) {
    if (g_ws[dev]) (void)hipFree(g_ws[dev]);
    size_t want = bytes < (32u << 20) ? (32u << 20) : bytes;
    if (hipMalloc(&g_ws[dev], want) != hipSuccess) {
      g_ws[dev] = nullptr;
      g_ws_bytes[dev] = 0;
      (void)hipGetLastError();
      return nullptr;
    }
    g_ws_bytes[dev] = want;
  }
  return g_ws[dev];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*gemm_fn)(const GemmArgs);

template <int KIND, int LAYOUT, int AT, int MODE, int FLAGS>
static gemm_fn pick_mf(int mf) {
  switch (mf) {
    case 16: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 16, 8>>;   // 256 x 256, 8 waves
    case 8: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 8>>;
    case 4: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 4>>;
    case 2: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 2>>;
    case 1: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1>>;
    // skinny members: 4 waves x 1 fragment, 4 k-steps per workgroup (mf code 100 + MF)
    case 101: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 1, 4, 1, 4>>;
    case 102: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 2, 4, 1, 4>>;
    case 104: return wq_gemm_kernel<GemmPolicy<KIND, LAYOUT, AT, MODE, FLAGS, 4, 4, 1, 4>>;
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

static gemm_fn pick_gemm(int kind, int layout, int at, int mode, int flags, int mf) {
  if (at == AT_F16) {
    switch (kind) {
      case DK_INT4: return layout == LAYOUT_LOP3 ? pick_modes<DK_INT4, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT4, LAYOUT_PLAIN>(mode, mf);
      case DK_INT2: return layout == LAYOUT_LOP3 ? pick_modes<DK_INT2, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT2, LAYOUT_PLAIN>(mode, mf);
      case DK_INT1: return layout == LAYOUT_LOP3 ? pick_modes<DK_INT1, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT1, LAYOUT_PLAIN>(mode, mf);
      case DK_INT8: return pick_modes<DK_INT8, LAYOUT_PLAIN>(mode, mf);
      case DK_LUT4: return pick_modes_fp<DK_LUT4, 0>(mode, mf);
      case DK_E4M3: return (flags & FL_STRICT) ? pick_modes_fp<DK_E4M3, FL_STRICT>(mode, mf) : pick_modes_fp<DK_E4M3, 0>(mode, mf);
      case DK_E5M2: return pick_modes_fp<DK_E5M2, 0>(mode, mf);
      case DK_NATIVE: return mode == MD_NONE ? pick_mf<DK_NATIVE, LAYOUT_PLAIN, AT_F16, MD_NONE, 0>(mf) : nullptr;
    }
    return nullptr;
  }
  if (mode != MD_NONE) return nullptr;
  if (at == AT_F8) {   // dense fp8 x fp8 (all four e4m3 / e5m2 pairings, general_matmul/__init__.py:33-51)
    const bool ab = (flags & FL_ABF8) != 0;
    if (kind == DK_E4M3) return ab ? pick_mf<DK_E4M3, LAYOUT_PLAIN, AT_F8, MD_NONE, FL_ABF8>(mf) : pick_mf<DK_E4M3, LAYOUT_PLAIN, AT_F8, MD_NONE, 0>(mf);
    if (kind == DK_E5M2) return ab ? pick_mf<DK_E5M2, LAYOUT_PLAIN, AT_F8, MD_NONE, FL_ABF8>(mf) : pick_mf<DK_E5M2, LAYOUT_PLAIN, AT_F8, MD_NONE, 0>(mf);
    return nullptr;
  }
  switch (kind) {
    case DK_INT4: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT4, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT4, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_INT2: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_INT1: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT1, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT1, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_NATIVE: return pick_mf<DK_NATIVE, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
  }
  return nullptr;
}

struct GemmChoice {
  gemm_fn fn;
  int kind, layout, at, mode, flags, bits;
  int mf, ks, kl, nwaves, bn, ksplit, skinny;
  int tiles_m, tiles_n, lds;
  int fp4_table;
};

static int gemm_choose(const wqaa_matmul_desc& d, int m, GemmChoice* c) {
  const int a = d.a_dtype;
  c->fp4_table = 0;
  c->flags = 0;
  c->layout = d.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN;
  if (a == WQAA_F16) c->at = AT_F16;
  else if (a == WQAA_I8) c->at = AT_I8;
  else if (a == WQAA_E4M3 || a == WQAA_E5M2) {
    c->at = AT_F8;
    if (a == WQAA_E5M2) c->flags |= FL_ABF8;
    if (d.with_bias) {
      set_error(WQAA_ERR_UNSUPPORTED, "gemm: fp8 x fp8 with bias is not defined by the reference");
      return WQAA_ERR_UNSUPPORTED;
    }
  } else {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: A dtype %d has no MFMA member yet", a);
    return WQAA_ERR_UNSUPPORTED;
  }
  c->bits = d.w_bits;
  switch (d.w_format) {
    case WQAA_W_UINT:
    case WQAA_W_INT:
      c->kind = d.w_bits == 4 ? DK_INT4 : d.w_bits == 2 ? DK_INT2 : d.w_bits == 1 ? DK_INT1 : d.w_bits == 8 ? DK_INT8 : -1;
      if (c->kind == DK_INT8 && c->at == AT_I8) c->kind = DK_NATIVE;
      break;
    case WQAA_W_NF: c->kind = d.w_bits == 4 ? DK_LUT4 : -1; break;
    case WQAA_W_FP4: c->kind = d.w_bits == 4 ? DK_LUT4 : -1; c->fp4_table = 1; break;
    case WQAA_W_E4M3: c->kind = DK_E4M3; break;
    case WQAA_W_E5M2: c->kind = DK_E5M2; break;
    case WQAA_W_NATIVE:
      c->kind = a == WQAA_E4M3 ? DK_E4M3 : a == WQAA_E5M2 ? DK_E5M2 : DK_NATIVE;
      c->bits = a == WQAA_F16 ? 16 : 8;
      break;
    default: c->kind = -1;
  }
  if (c->kind < 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: weight format %d / %d bits not supported", d.w_format, d.w_bits);
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->at == AT_F8 && c->kind != DK_E4M3 && c->kind != DK_E5M2) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: fp8 activations need fp8 weights");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->kind == DK_E4M3 && d.strict_reference && c->at == AT_F16) c->flags |= FL_STRICT;
  if (c->kind != DK_INT4 && c->kind != DK_INT2 && c->kind != DK_INT1) c->layout = LAYOUT_PLAIN;
  if (c->at != AT_F16 && (d.with_scaling || d.zeros_mode != WQAA_Z_NONE)) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: scale/zeros with int8 / fp8 activations are not defined by the reference");
    return WQAA_ERR_UNSUPPORTED;
  }
  c->mode = !d.with_scaling ? MD_NONE
            : d.zeros_mode == WQAA_Z_ORIGINAL ? MD_ZO
            : d.zeros_mode == WQAA_Z_RESCALE  ? MD_ZR
            : d.zeros_mode == WQAA_Z_QUANTIZED ? MD_ZQ
                                               : MD_S;
  c->kl = c->at == AT_F16 ? 32 : 64;   // int8: 16 k per MFMA lane; fp8: 8 k per MFMA lane, two MFMAs per granule
  c->ks = 4 * c->kl;
  if (d.K % c->ks != 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: K=%d must be a multiple of %d", d.K, c->ks);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  if (d.K % g != 0 || (c->mode != MD_NONE && g % c->kl != 0)) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: group_size=%d must divide K=%d and be a multiple of %d", g, d.K, c->kl);
    return WQAA_ERR_UNSUPPORTED;
  }
  if (d.N % 4 != 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: N=%d must be a multiple of 4", d.N);
    return WQAA_ERR_UNSUPPORTED;
  }
  // BM: the largest tile that M fills; small M keeps more workgroups alive along N
  // BM (measured sweep, tools/sweep_gemm.sh, N = K = 4096): the 256 x 256 / 8-wave tile wins once it
  // gives every CU a workgroup; below that 128-row tiles (+ split-K) keep more workgroups alive, and
  // skinny M follows M down
  const int cus_ = device_info().ok ? device_info().cus : 256;
  const long tiles256 = (long)((m + 255) / 256) * ((d.N + 255) / 256);
  c->mf = (m >= 256 && d.N >= 256 && tiles256 * 10 >= (long)cus_ * 9) ? 16
          : m > 128 ? 8 : m > 32 ? 4 : m > 16 ? 2 : 1;
  if (const char* f = getenv("WQAA_GEMM_MF")) c->mf = atoi(f);   // tuning aid
  const int nsteps = d.K / c->ks;
  // decode batches (M <= 64): the skinny member, unless disabled
  c->skinny = (m <= 64 && c->mf <= 4 && getenv("WQAA_GEMM_NOSKINNY") == nullptr) ? 4 : 0;
  c->nwaves = c->mf == 16 ? 8 : 4;
  c->bn = c->skinny ? 64 : c->nwaves * 32;
  c->fn = pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, c->skinny ? 100 + c->mf : c->mf);
  if (!c->fn) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: no kernel for kind=%d layout=%d at=%d mode=%d", c->kind, c->layout, c->at, c->mode);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int bm = 16 * c->mf, bn = c->bn;
  c->tiles_m = (m + bm - 1) / bm;
  c->tiles_n = (d.N + bn - 1) / bn;
  c->lds = (c->skinny ? c->skinny : 2) * bm * 256;
  if (c->skinny) {
    c->ksplit = (nsteps + c->skinny - 1) / c->skinny;
    return WQAA_OK;
  }
  // split-K: a skinny problem has too few tiles to fill 256 CUs; cut K until ~1 workgroup per CU
  // (partials cost 4 B per output element per slice, so stop at 16)
  const int cus = device_info().ok ? device_info().cus : 256;
  const int tiles = c->tiles_m * c->tiles_n;
  int ks = 1;
  while (tiles * ks < cus && ks * 2 <= 16 && nsteps / (ks * 2) >= 1) ks *= 2;
  if (const char* f = getenv("WQAA_GEMM_KSPLIT")) ks = atoi(f);
  if (ks > nsteps) ks = nsteps;
  if (ks < 1) ks = 1;
  c->ksplit = ks;
  return WQAA_OK;
}

int gemm_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan) {
  GemmChoice c;
  int st = gemm_choose(d, m, &c);
  if (st != WQAA_OK) return st;
  if (plan) {
    plan->kernel_family = 2;
    plan->block_m = 16 * c.mf;
    plan->block_n = c.bn;
    plan->block_k = c.ks;
    plan->threads = 64 * c.nwaves;
    plan->grid = c.tiles_m * c.tiles_n;
    plan->rows_per_wave = 32;
    plan->batch_tile = 16 * c.mf;
    plan->pipeline_depth = 2;
    plan->split_k = c.ksplit;
    plan->lds_bytes = c.lds;
    plan->grid = c.tiles_m * c.tiles_n * c.ksplit;
    snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_a%dw%db%d_tcx%dx%dx%d%s", m, d.N, d.K, d.a_dtype,
             d.w_format, d.w_bits, 16 * c.mf, c.bn, c.ks, c.ksplit > 1 ? "xr" : "");
  }
  return WQAA_OK;
}

int gemm_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  GemmChoice c;
  int st = gemm_choose(d, m, &c);
  if (st != WQAA_OK) return st;
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  GemmArgs a;
  a.A = A; a.B = B; a.lut = LUT; a.scale = Scale; a.zeros = Zeros; a.bias = Bias; a.C = C;
  a.M = m; a.N = d.N; a.K = d.K;
  a.kg = d.K / g;
  {
    const int dq = g / c.kl > 0 ? g / c.kl : 1;
    a.gq_shift = ilog2_exact(dq);
    a.gq_magic = a.gq_shift >= 0 ? 0u : (uint32_t)(((1ull << 32) + dq - 1) / dq);
  }
  a.row_bytes = (long)d.K * c.bits / 8;
  a.has_bias = d.with_bias;
  a.out_dtype = d.out_dtype;
  a.is_signed = d.w_format == WQAA_W_INT;
  a.fp4_table = c.fp4_table;
  a.zq_row_bytes = d.N * (c.bits < 8 ? c.bits : 8) / 8;
  a.tiles_m = c.tiles_m;
  a.tiles_n = c.tiles_n;
  a.nsteps = d.K / c.ks;
  a.ksplit = c.ksplit;
  a.ws = nullptr;
  if (c.ksplit > 1) {
    a.ws = workspace((size_t)c.ksplit * m * d.N * 4);
    if (!a.ws) {
      set_error(WQAA_ERR_LAUNCH, "gemm: cannot allocate %zu B of split-K scratch", (size_t)c.ksplit * m * d.N * 4);
      return WQAA_ERR_LAUNCH;
    }
  }
  void* params[] = {&a};
  dim3 grid(c.tiles_m * c.tiles_n * c.ksplit, 1, 1), block(64 * c.nwaves, 1, 1);
  hipError_t e;
  if (start || stop) {
    e = hipExtLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream, start,
                           c.ksplit > 1 ? nullptr : stop, 0);
  } else {
    e = hipLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream);
  }
  if (e == hipSuccess && c.ksplit > 1) {
    const long quads = (long)m * d.N / 4;
    const dim3 rgrid((unsigned)((quads + 255) / 256)), rblock(256);
    const void* ws = a.ws;
    int M_ = m, N_ = d.N, ks_ = c.ksplit, od = d.out_dtype, hb = d.with_bias;
    void* rparams[] = {&ws, &C, &Bias, &M_, &N_, &ks_, &od, &hb};
    const void* rfn = c.at != AT_I8 ? reinterpret_cast<const void*>(wq_splitk_reduce_kernel<true>)
                                     : reinterpret_cast<const void*>(wq_splitk_reduce_kernel<false>);
    if (start || stop) e = hipExtLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream, nullptr, stop, 0);
    else e = hipLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream);
  }
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "gemm launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

void gemm_init() {
  const int kinds[] = {DK_INT4, DK_INT2, DK_INT1, DK_INT8, DK_LUT4, DK_E4M3, DK_E5M2, DK_NATIVE};
  for (int kind : kinds)
    for (int layout = 0; layout < 2; ++layout)
      for (int at = 0; at < 3; ++at)
        for (int mode = 0; mode <= MD_ZQ; ++mode)
          for (int flags : {0, (int)FL_STRICT, (int)FL_ABF8})
            for (int mf : {1, 2, 4, 8, 16, 101, 102, 104}) {
              gemm_fn fn = pick_gemm(kind, layout, at, mode, flags, mf);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
}

}  // namespace wqaa
