// wqaa_gemm_pp_kernel.h - the large-M W_q x A MFMA member: a role-alternating ("ping-pong") main loop for gfx950.
//
// Replaces the reference's tensor-core main loops `MatmulDequantizeMMAScheduler` / `MatmulMMAScheduler`
// (bitblas/ops/general_matmul/tilelang/dequantize/matmul_dequantize_mma.py:333-508, tilelang/dense/matmul_mma.py:220-320)
// for M >= 256.  Same computation and the same per-element dequant arithmetic as wq_gemm_kernel (wqaa_gemm_kernel.h);
// what changes is the machine schedule:
//   * 256 x 256 tile (128 x 256 where that fills the chip in fewer rounds), 8 waves, every wave owns 32 weight rows (n) and
//     multiplies them by all the tile's activation rows.  The matrix instruction is the 16x16 form (v_mfma_f32_16x16x32_f16 /
//     v_mfma_i32_16x16x64_i8), weights as its A operand: a lane owns weight row n = lane & 15 of a 16-row fragment and the
//     k-block lane >> 4, so ONE 32-bit word of packed weights (8 int4 / 16 int2) decodes in registers into exactly one MFMA
//     operand.  Packed weights never become fp16 in LDS.  (The 32x32 form halves the LDS reads per flop but draws more power
//     per flop - tools/mfma_power.hip: 1774 vs 1950 TFLOP/s sustained on random operands - and the chip is power-limited.)
//   * every byte from global memory arrives by LDS-DMA (buffer_load ... lds): the activation tile (256 rows x 128 B per
//     k-tile, XOR-swizzled through the SOURCE address so the ds_read_b128 operand reads are conflict free), the wave's
//     own packed weights (1 KiB per k-tile, read back by the lane that owns them) and its Scale / Zeros.  One load
//     KIND in flight means the counted `s_waitcnt vmcnt(N)` is in order, so the prefetch ring (RING k-tiles) is never
//     drained inside the loop, and no VGPR waits for memory;
//   * the two waves of a SIMD alternate roles (MI355X_MICROARCH "Two waves per SIMD"): waves 0-3 and waves 4-7 run the
//     same instruction stream one s_barrier apart, so while one wave of a SIMD is in a COMPUTE segment (16 MFMAs with the
//     decode of the next weight word interleaved in their shadow), its partner is in a LOAD segment (8 ds_read_b128 for
//     its next 16 MFMAs, its share of the LDS-DMA issue, the waits).  No register double buffer for the operands: the
//     partner's compute covers the LDS latency;
//   * the output tile leaves through LDS, whole rows per store instruction (the accumulator layout gives a lane 4
//     consecutive n of one m: 8-byte pieces 512 B apart; the store tail was issue-bound).
#pragma once
#include "wqaa_gemm_kernel.h"

namespace wqaa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

enum : int {
  PPO_ZINT_OFF = 1,      // never take the integer-zero-point decode (A/B aid)
  PPO_RW64 = 2,          // lab: the packed words come back by conflict-free 8-byte reads + a lane select (4-byte reads: 2-way conflicts)
  PPO_META_SLOW = 4,     // lab: the general Scale / Zeros window arithmetic even where the rows' windows are aligned
  PPO_TRACE = 32,        // lab: s_memtime stamps of the four phases of k-tile 16, written through a.lut ([block][wave][20])
  PPO_ABL_NODMA = 64,    // lab ablations (timing only, results wrong): no LDS-DMA inside the loop,
  PPO_ABL_NOREAD = 128,  //   no operand reads from LDS,
  PPO_ABL_NODEC = 256,   //   no weight decode
  PPO_ABL_NOBAR = 512,   //   no s_barrier in the loop (wqaa_gemm_mm_kernel.h),
  PPO_ABL_NOMFMA = 1024, //   no MFMA (wqaa_gemm_mm_kernel.h)
  PPO_PRIO = 4096,       // lab: s_setprio 1 for the second-dispatched role group (waves 4-7) across the main loop
  PPO_ABL_METAONCE = 2048, // Scale / Zeros read and converted for the first k-body only (what the per-body metadata handling costs)
  PPO_ABL_LONGSEG = 8192,  // lab: two phases per barrier interval - 32 MFMAs per compute segment, half the barriers, 32 more registers (results unchanged)
};

// BM_ = 256: the full tile.  BM_ = 128: the same loop on half the activation rows, for shapes whose 256-row tiles would leave the
// chip short of whole rounds (M = 512 ... 3584 at N = 4096 ... 22016): two phases per k-tile instead of four, so a k-tile
// lasts half as long and the latencies are covered by depth instead - ring of 5 k-tiles (the slots are half the size) and the
// weight chunks double buffered, asked for a whole trip ahead.
template <int KIND_, int LAYOUT_, int AT_, int MODE_, int FLAGS_, int RING_ = 3, int OPT_ = 0, int BM_ = 256>
struct PPPolicy {
  static constexpr int KIND = KIND_, LAYOUT = LAYOUT_, AT = AT_, MODE = MODE_, FLAGS = FLAGS_, RING = RING_, OPT = OPT_;
  static constexpr int D = RING_ - 1;               // k-tiles of prefetch distance
  static constexpr int BM = BM_, BN = 256, THREADS = 512, NWAVES = 8;
  static constexpr int NPH = BM_ / 64;              // phases (16 MFMAs per wave each) per k-tile: 4 / 2
  static constexpr int WBUFS = BM_ == 128 ? 2 : 1;  // weight-chunk buffers per wave
    static constexpr int TILE_ROW = 128;              // bytes of one activation row per k-tile
  static constexpr int KT = AT_ == AT_F16 ? 64 : 128;   // k per tile
  static constexpr int KB = 4 * KT;                 // k per loop trip: four k-tiles = one 128-byte line of every weight row
  static constexpr int A_SLOT = BM * TILE_ROW;      // 32 / 16 KiB
  static constexpr int W_OFF = RING_ * A_SLOT;      // per wave: WBUFS 4 KiB chunks (32 rows x 128 B = four k-tiles)
  static constexpr int META_OFF = W_OFF + NWAVES * WBUFS * 4096;
  static constexpr bool HAS_META = MODE_ != MD_NONE;
  static constexpr int LDS_USED = META_OFF + (HAS_META ? NWAVES * 2 * 1024 : 0);   // per wave: two 1 KiB windows (8 groups of Scale | Zeros)
  static constexpr int EPI_BYTES = AT_ == AT_F16 ? BM * BN * 2 : 128 * BN * 4;        // the epilogue stages the output tile (int32: 128 rows a pass)
  static constexpr int LDS_BYTES = LDS_USED > EPI_BYTES ? LDS_USED : EPI_BYTES;
  using T = KindTraits<KIND_, AT_>;
  static_assert(T::BITS * KT == 256, "one k-tile of a weight row is 32 bytes: 16 per lane half");
  static_assert(BM_ == 256 || BM_ == 128, "256- or 128-row tile");
  static_assert(BM_ == 128 ? RING_ == 5 : (RING_ == 3 || RING_ == 4), "ring of 3 or 4 k-tiles (5 for the 128-row tile: its counted wait assumes it)");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int V>
using ic = std::integral_constant<int, V>;

#define PP_FENCE() asm volatile("" ::: "memory")
#define PP_BARRIER()                        \
  do {                                      \
    PP_FENCE();                             \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
    PP_FENCE();                             \
  } while (0)

// a value the optimiser cannot trace back: what is computed from it stays where it is written (rarely used addresses are
// recomputed at their use instead of occupying registers across the main loop)
__device__ __forceinline__ int pp_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// 16 bytes of the output tile.  A large output (GemmArgs::ws_policy bit 4, set by the launcher from M x N) is stored
// write-through: left dirty in L2 it is written back at the kernel boundary, in front of whatever runs next
// (MI355X_MICROARCH "boundary": + B / 6 TB/s behind B dirty bytes; same-process A/B profiles/r03_ab_out_policy.txt: -2 %)
template <class V>
__device__ __forceinline__ void pp_store_out(V* dst, const V& x, int policy) {
  if (policy & 16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(x) : "memory");
  else *dst = x;
}

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one packed word -> one MFMA operand (8 fp16 in natural k order); arithmetic identical to dequant_lane_f16.
// ZINT (zeros-original, every zero point of the wave's rows an integer the magic exponents hold exactly): zA / zB are
// (2^10 + zf + z) / (2^6 + zf + z), and (magic + q) - (magic + z) IS q - z, exactly - one vector operation less per pair.
template <class P, bool ZINT>
__device__ __forceinline__ void pp_decode_f16(uint32_t w, half_t zf, half2_t s2, half2_t zA, half2_t zB, const DecodeCtx& cx, const Lut16& lut,
                                              uint32_t (&out)[4]) {
  using T = typename P::T;
  static_assert(T::EPW == 8, "4-bit weights");
  half2_t q[4];
  if constexpr (P::KIND == DK_LUT4) {
    lut16_word(lut, w, q);
  } else if constexpr (ZINT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int bit = 4 * i, b = bit & 7;
      const uint32_t src = (bit >= 8) ? (w >> 8) : w;
      const uint32_t m = (0xFu << b) * 0x00010001u;
      const uint32_t t = (src & m) | cx.magic[b];
      q[i] = as_h2(t) - (b ? zB : zA);
    }
  } else {
    F16Unpack<4>::run(w, zf, cx.magic, q);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (P::MODE == MD_S || P::MODE == MD_ZQ) q[i] = q[i] * s2;
    if constexpr (P::MODE == MD_ZO) {
      if constexpr (ZINT) q[i] = q[i] * s2;
      else q[i] = (q[i] - zA) * s2;
    }
    if constexpr (P::MODE == MD_ZR) {
      half2_t t = q[i] * s2;
      asm volatile("" : "+v"(t));   // two roundings, no fma contraction
      q[i] = t - zA;
    }
  }
  to_natural_f16<T, P::LAYOUT>(q, out, std::make_integer_sequence<int, 4>{});
}

// bfloat16 activations (FL_BF16; plain layout): the lockstep member's decode, one word at a time (unpack_word_bf16: field ->
// float -> the mode's arithmetic with its bfloat16 roundings -> packed bfloat16 pairs in natural k order)
template <class P>
__device__ __forceinline__ void pp_decode_bf16(uint32_t w, float zf, float s, float z, const Lut16& lut, uint32_t (&out)[4]) {
  if constexpr (P::KIND == DK_LUT4) {
    // nf4 / fp4: the table holds bfloat16 bit patterns; the scale is applied with one rounding, then natural order
    using T = typename P::T;
    half2_t q[4];
    lut16_word(lut, w, q);
    if constexpr (P::MODE != MD_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = as_h2(bf16x2_scale(as_u32(q[i]), s));
    }
    to_natural_f16<T, P::LAYOUT>(q, out, std::make_integer_sequence<int, 4>{});
  } else {
    constexpr int ZM = P::MODE == MD_ZO ? 1 : P::MODE == MD_ZR ? 2 : 0;
    unpack_word_bf16<4, 1, ZM>(w, zf, s, P::MODE != MD_NONE, out, z);
  }
}

template <class P>
__device__ __forceinline__ void pp_decode_i8(uint32_t w, uint32_t zp4, uint32_t flip, uint32_t (&out)[4]) {
  using T = typename P::T;
  static_assert(T::EPW == 16, "2-bit weights");
  uint32_t t[4];
  I8Unpack<2>::run(w ^ flip, t);
  const uint32_t tbl = sub_bytes(0x03020100u, zp4);      // (loop invariant: the zero point is one value per launch)
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = sub_bytes_tbl(t[i], tbl);
  to_natural_i8<T, P::LAYOUT>(t, out, std::make_integer_sequence<int, 4>{});
}

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_pp_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the stub (the buffer-resource builtins do not exist there)
  using T = typename P::T;
  constexpr bool F16 = P::AT == AT_F16;
  constexpr bool BF = F16 && (P::FLAGS & FL_BF16) != 0;     // bfloat16 activations / Scale / Zeros / output (4-bit integer weights, plain layout)
  constexpr int MODE = P::MODE, D = P::D, RING = P::RING;
  constexpr int NMF = P::BM / 16;       // 16-row activation fragments: 16 / 8
  constexpr int NPH = P::NPH, HP = NPH / 2;   // phases per k-tile; phases per MFMA (k-half) of the tile
  constexpr int WROWS = P::BM / 8;      // activation rows a wave feeds per k-tile: 32 / 16 (NPH pieces of 8)
  constexpr bool HALF = P::BM == 128;
  constexpr bool ZP = MODE == MD_ZO || MODE == MD_ZR;
  constexpr bool ZQ = MODE == MD_ZQ;    // GPTQ-style packed integer zero points: Zeros[K / g][N / 2] bytes, nibble n & 1 of byte n / 2
  using acc_t = typename std::conditional<F16, f32x4, i32x4>::type;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;            // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
  const int ln = lane & 31, h = lane >> 5;   // LDS-DMA / metadata role of the lane: row ln of the wave's 32, half h
  const int fr = lane & 15, kb = lane >> 4;  // MFMA role: fragment row (weight n / activation m), k-block

  unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_c0 = 0, tr_r0 = 0;
  if constexpr (P::OPT & PPO_TRACE) {
    tr_c0 = __builtin_readcyclecounter();
    tr_r0 = __builtin_amdgcn_s_memrealtime();
  }
  auto stamp = [&](int t, int idx) {
    if constexpr (P::OPT & PPO_TRACE) {
      if (t == 16) tr[idx] = __builtin_readcyclecounter();
    }
  };

  // ---- tile of this workgroup: XCD-contiguous ranges of the grouped order (see wq_gemm_kernel) ----
  const TileOfBlock tob = tile_of_block(a, (int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tob.tile_m, tile_n = tob.tile_n;
  const int m0 = tile_m * P::BM;
  const int n0 = (tile_n + a.tile_n_off) * P::BN;
  const int nw0 = n0 + wave * 32;       // first weight row of this wave

  const int ntiles = a.K / P::KT;       // a multiple of 4 (K is a multiple of KB)
  const int nchunks = ntiles >> 2;

  // ---- LDS-DMA sources ----
  const int a_row_bytes = F16 ? a.K * 2 : a.K;
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((long)a.M * a_row_bytes), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.B), 0, (int)((long)a.N * a.row_bytes), 0x00020000);
  // activation piece j of a tile: this wave's rows [wave * 32 + j * 8, + 8), 8 lanes per row (one 128-byte line); the lane
  // that fills physical granule p of row r fetches natural granule p ^ ((r >> 1) & 7) = p ^ (((j & 1) * 4 + (lane >> 4)) & 7).
  // Two registers serve the four pieces: the offset of piece 0 and the +-64 bytes an odd piece's swizzle moves the granule by;
  // rows and k-tile go through scalar adds.  Rows beyond M are out of the buffer's range (they read as zero, never stored).
  const int g0_ = (lane & 7) ^ ((lane >> 4) & 7);
  const uint32_t a_v0 = (uint32_t)(wave * WROWS + (lane >> 3)) * (uint32_t)a_row_bytes + (uint32_t)(g0_ * 16);
  const int a_vd = ((g0_ ^ 4) - g0_) * 16;
  const uint32_t a_rows0 = (uint32_t)m0 * (uint32_t)a_row_bytes;
  // weight piece p of a chunk: rows [8p, 8p + 8) of this wave's 32, the 128-byte line that holds four k-tiles of the row,
  // swizzled like the activations (offsets recomputed at every use: four uses per trip of the main loop)
  auto w_voff = [&](int p) -> uint32_t {
    const int l = pp_opaque(lane);
    const int r = p * 8 + (l >> 3);
    const int g = (l & 7) ^ ((r >> 1) & 7);
    const int row = nw0 + r < a.N ? nw0 + r : a.N - 1;
    return (uint32_t)row * (uint32_t)a.row_bytes + (uint32_t)(g * 16);
  };
  const int nrow = nw0 + ln < a.N ? nw0 + ln : a.N - 1;
  // Scale (lanes 0-31) / Zeros (lanes 32-63) of row nrow come as 16-byte windows of 8 consecutive groups and are consumed by
  // the lanes that own rows fr and 16 + fr of the MFMA operand
  const uint32_t mlim = P::HAS_META ? (uint32_t)a.N * (uint32_t)a.kg - 8u : 0u;
  const uint32_t rowbase = (uint32_t)nrow * (uint32_t)a.kg;     // (prologue only)
  const int nbodies = ntiles >> 1;
  auto group_of_body = [&](int b) -> int {   // k-body b (two k-tiles: 128 k for fp16) -> group index; a group is 2^gq_shift bodies
    b = b < nbodies ? b : nbodies - 1;
    return b >> a.gq_shift;
  };
  // first element of window q of a row: kept inside the array and on a 4-byte boundary (a row starts on one whenever K / g is
  // even; with ONE group per row - per-channel scales - an odd row's window opens one element early)
  auto window_start = [&](uint32_t rb, int q) -> uint32_t {
    const uint32_t e = (rb + (uint32_t)(q * 8)) & ~1u;
    return e < mlim ? e : mlim;
  };

  unsigned char* const a_ring = smem;
  unsigned char* const w_buf = smem + P::W_OFF + wave * (P::WBUFS * 4096);
  unsigned char* const meta = smem + P::META_OFF + wave * 2048;

  auto dma_a = [&](int tt, int slot, int j) {       // activation piece j of k-tile tt -> ring slot
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = a_ring + slot * P::A_SLOT + (wave * WROWS + j * 8) * P::TILE_ROW;
    const uint32_t rows = a_rows0 + (uint32_t)(j * 8) * (uint32_t)a_row_bytes;     // scalar
    const uint32_t voff = (j & 1) ? a_v0 + (uint32_t)a_vd + rows : a_v0 + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };
  auto dma_w = [&](int chunk, int p, int buf) {
    const int cc = chunk < nchunks ? chunk : nchunks - 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(w_buf + buf * 4096 + p * 1024), 16, w_voff(p), cc * 128, 0, 0);
  };
  auto dma_meta = [&](int q) {                      // window q -> buffer q & 1
    if constexpr (P::HAS_META) {
      const int l = pp_opaque(lane);
      const int n = nw0 + (l & 31);
      if constexpr (ZQ) {
        // lanes 0-31: the Scale windows; lanes 32-39: the 16 bytes that hold the zero points of the wave's 32 rows, one lane per
        // group of the window (N is a multiple of 32 for this member: a wave's rows are all inside the matrix or all outside)
        if (l < 40) {
          const int gz = q * 8 + (l & 7) < a.kg ? q * 8 + (l & 7) : a.kg - 1;
          const int nz = nw0 + 32 <= a.N ? nw0 : a.N - 32;
          const unsigned char* src = l < 32
              ? reinterpret_cast<const unsigned char*>(reinterpret_cast<const uint16_t*>(a.scale) + window_start((uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg, q))
              : reinterpret_cast<const unsigned char*>(a.zeros) + (long)gz * a.zq_row_bytes + (nz >> 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (lds_ptr_t)(meta + (q & 1) * 1024), 16, 0, 0);
        }
      } else {
        const uint16_t* mbase = (ZP && (l >> 5) == 1) ? reinterpret_cast<const uint16_t*>(a.zeros) : reinterpret_cast<const uint16_t*>(a.scale);
        const uint16_t* src = mbase + window_start((uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg, q);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (lds_ptr_t)(meta + (q & 1) * 1024), 16, 0, 0);
      }
    }
  };

  // ---- operand reads ----
  // A k-tile is 8 granules of 16 bytes per row; MFMA jj (0, 1) of the tile takes granule 4 * jj + kb from the lane: the two
  // 8-lane halves of a ds_read_b128 service group (k-blocks kb, kb ^ 1) then never meet on a 16-byte slot
  const int swl = (fr >> 1) & 7;
  uint32_t a_rd[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) a_rd[jj] = (uint32_t)(fr * P::TILE_ROW + (((4 * jj + kb) ^ swl) * 16));
  DecodeCtx cx;
  cx.zf = (F16 && a.is_signed && P::KIND != DK_LUT4 && !ZQ) ? (half_t)8.0f : (half_t)0.0f;   // (quantized zeros act in the code domain: no sign offset)
  cx.flip = 0u;
  cx.off8 = (half_t)0.0f;
  if constexpr (F16) {                     // 4-bit fields sit at bit 0 and bit 4 of a byte: two magic exponent words, pinned in VGPRs
#pragma unroll
    for (int b = 0; b < 8; ++b) cx.magic[b] = (uint32_t)((25 - b) << 10) * 0x00010001u;
    asm volatile("" : "+v"(cx.magic[0]));
    asm volatile("" : "+v"(cx.magic[4]));
  }
  const uint32_t zp4 = (!F16 && a.is_signed) ? 0x02020202u : 0u;
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) lut = make_fp4_lut(BF);
    else lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
  }

  bool zint = ZQ;                         // packed integer zero points always take the magic-exponent decode

  acc_t acc[NMF][2];
#pragma unroll
  for (int f = 0; f < NMF; ++f)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) acc[f][nf] = acc_t{0, 0, 0, 0};

  // (LONGSEG, lab: the odd phases have their own fragment registers - two phases are loaded, then two computed)
  constexpr int AF2 = (P::OPT & PPO_ABL_LONGSEG) ? 8 : 0;
  u32x4 afrag[8 + AF2];
  uint32_t bw[2][2][4];                   // decoded weight operands: [pair parity][n fragment]
  uint32_t rawc[4][2][2];                 // packed words of the chunk in hand, read half a chunk at a time: [k-tile][n fragment][MFMA of the tile]
  half2_t s2c[2], zAc[2], zBc[2];         // Scale / Zeros of the k-body being decoded, per weight fragment
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    s2c[nf] = splat((half_t)1.0f);
    zAc[nf] = zBc[nf] = splat((half_t)0.0f);
  }
  uint32_t m_s[2] = {0, 0}, m_z[2] = {0, 0};   // Scale / Zeros bits of the next body

  // element of a row's window that holds group gi: gi - min(8 q, mlim - rowbase); the per-row term is kept per weight fragment
  int mlim_f[2] = {0, 0};
  if constexpr (P::HAS_META) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const int n = nw0 + nf * 16 + fr;
      mlim_f[nf] = (int)(mlim - (uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg);
    }
  }
  auto meta_read = [&](int b) {            // Scale / Zeros of body b (its window has landed)
    if constexpr ((P::OPT & PPO_ABL_METAONCE) != 0) {
      if (b != 0) return;
    }
    if constexpr (P::HAS_META) {
      const int gi = group_of_body(b);
      const int q8 = gi & ~7;
      const unsigned char* p = meta + ((gi >> 3) & 1) * 1024 + (pp_opaque(lane) & 15) * 16;
      // e = gi - min(8 q - parity, mlim_f) = max((gi & 7) + parity, gi - mlim_f): the scalar part of each term stays scalar
      // (every vector operation of this loop shows in its time: the ~120 that Scale + Zeros add per trip are its 15 us)
      const int c1 = (gi & 7) + (mlim_f[0] & 1);    // (mlim is even: the parity of mlim - rowbase is the row's; both fragments' rows share it - 16 rows apart)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        int e;
        if constexpr (P::OPT & PPO_META_SLOW) {
          const int q8o = q8 - (mlim_f[nf] & 1);
          e = gi - (q8o < mlim_f[nf] ? q8o : mlim_f[nf]);
        } else {
          const int t = gi - mlim_f[nf];
          e = t > c1 ? t : c1;
        }
        m_s[nf] = *reinterpret_cast<const uint16_t*>(p + nf * 256 + e * 2);
        if constexpr (ZP) m_z[nf] = *reinterpret_cast<const uint16_t*>(p + 512 + nf * 256 + e * 2);
        if constexpr (ZQ) {
          const int r = nf * 16 + (pp_opaque(lane) & 15);
          const uint32_t b = *reinterpret_cast<const uint8_t*>(meta + ((gi >> 3) & 1) * 1024 + 512 + (gi & 7) * 16 + (r >> 1));
          m_z[nf] = (b >> ((r & 1) * 4)) & 15u;
        }
      }
    }
  };
  auto meta_convert = [&](auto ZI, half2_t (&s2)[2], half2_t (&zA)[2], half2_t (&zB)[2]) {
    if constexpr (P::HAS_META && BF) {
      // bfloat16: Scale and zero point travel as fp32 in the same registers
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        s2[nf] = __builtin_bit_cast(half2_t, bf16_bits_to_float(m_s[nf]));
        if constexpr (ZP) zA[nf] = __builtin_bit_cast(half2_t, bf16_bits_to_float(m_z[nf]));
        if constexpr (ZQ) zA[nf] = __builtin_bit_cast(half2_t, (float)m_z[nf]);
      }
    } else if constexpr (P::HAS_META) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        s2[nf] = splat(bits_to_half(m_s[nf]));
        if constexpr (ZP) {
          const half_t z = bits_to_half(m_z[nf]);
          if constexpr (decltype(ZI)::value) {
            zA[nf] = splat((half_t)1024.0f + cx.zf + z);
            zB[nf] = splat((half_t)64.0f + cx.zf + z);
          } else {
            zA[nf] = splat(z);
          }
        }
        if constexpr (ZQ) {               // an integer 0..15: always the magic-exponent form
          const half_t z = (half_t)(float)m_z[nf];
          zA[nf] = splat((half_t)1024.0f + z);
          zB[nf] = splat((half_t)64.0f + z);
        }
      }
    }
  };
  auto decode = [&](auto ZI, uint32_t w, half2_t s2, half2_t zA, half2_t zB, uint32_t (&out)[4]) {
    if constexpr (BF) {
      // (quantized zeros: the integer zero point replaces the sign offset, as in the lockstep member)
      const float zv = __builtin_bit_cast(float, zA);
      pp_decode_bf16<P>(w, ZQ ? zv : (float)cx.zf, __builtin_bit_cast(float, s2), zv, lut, out);
    } else if constexpr (F16) pp_decode_f16<P, decltype(ZI)::value != 0>(w, cx.zf, s2, zA, zB, cx, lut, out);
    else pp_decode_i8<P>(w, zp4, cx.flip, out);
  };
  auto read_words = [&](int half, int buf) {        // k-tiles 2 * half, 2 * half + 1 of the landed chunk (in buffer buf) -> registers
    const int l = pp_opaque(lane);
    const int swl = ((l & 15) >> 1) & 7;
    if constexpr (P::OPT & PPO_RW64) {
      // a 4-byte read is banked modulo 32 dwords over lanes 0-31 / 32-63: rows 2j and 2j + 1 share a swizzle and meet on a bank
      // (2-way: the 8 % of LDS cycles SQ_LDS_BANK_CONFLICT counts for this member).  8-byte reads are banked modulo 64: the
      // 16 rows of a lane half fall on 16 different slots.  A lane reads the 8-byte half that holds its word and keeps one.
      typedef __attribute__((address_space(3))) const volatile uint64_t lds_u64;
      const uint32_t base = (uint32_t)(uintptr_t)(w_buf + buf * 4096) + (uint32_t)((l & 15) * 128 + (l >> 5) * 8);
      const bool hi = (l & 16) != 0;
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const uint64_t v = *reinterpret_cast<lds_u64*>(base + (uint32_t)(nf * 2048) + (uint32_t)(((2 * (2 * half + tp) + jj) ^ swl) * 16));
            rawc[2 * half + tp][nf][jj] = hi ? (uint32_t)(v >> 32) : (uint32_t)v;
          }
      return;
    }
    const uint32_t w_rd0 = (uint32_t)((l & 15) * 128 + (l >> 4) * 4);
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          rawc[2 * half + tp][nf][jj] = *reinterpret_cast<const uint32_t*>(w_buf + buf * 4096 + nf * 2048 + w_rd0 + (uint32_t)(((2 * (2 * half + tp) + jj) ^ swl) * 16));
  };

  // ---- prologue: the first window, chunk 0 and D k-tiles in flight; tile 0 landed; first operands decoded ----
  dma_meta(0);
#pragma unroll
  for (int b = 0; b < P::WBUFS; ++b)
#pragma unroll
    for (int p = 0; p < 4; ++p) dma_w(b, p, b);
#pragma unroll
  for (int tt = 0; tt < D; ++tt)
#pragma unroll
    for (int j = 0; j < NPH; ++j) dma_a(tt, tt, j);
  // zeros-original: are all zero points of this wave's rows integers the magic subtraction holds exactly?  (asked for behind
  // the first LDS-DMA pieces: the answer travels with them)
  if constexpr (F16 && !BF && MODE == MD_ZO && P::KIND == DK_INT4 && !(P::OPT & PPO_ZINT_OFF)) {
    bool ok = true;
    const uint16_t* zrow = reinterpret_cast<const uint16_t*>(a.zeros);
    for (int i = 0; i < a.kg; i += 8) {
      const uint32_t e = rowbase + (uint32_t)i;
      const u32x4 v = *reinterpret_cast<const u32x4*>(zrow + ((e < mlim ? e : mlim) & ~1u));   // (neighbours' elements checked too - near the end, and in front of an odd one-group row: conservative)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = (float)bits_to_half(v[k >> 1] >> ((k & 1) * 16)) + (float)cx.zf;
        ok = ok && z == __builtin_truncf(z) && z > -48.f && z < 48.f;
      }
    }
    zint = __all(ok);
  }
  // 256-row tile: tiles 1 .. D-1 may stay in flight - the loop's counted wait is exact from tile 0 on (the prologue ends with the activation
  // tiles).  128-row tile (D = 4): the loop's `vmcnt(10)` counts the ten pieces that FOLLOW tile t + 1 in steady state (three tiles of two
  // activation pieces + one chunk of four weight pieces); behind the prologue only six (tile 0) and six (tile 1) follow, so that wait let
  // tiles 1 and 2 be read before they had landed whenever their pieces took longer than tile 0's by a few hundred ns - seen as a handful
  // of wrong outputs about once in 2000 FIRST launches on fresh buffers (round 6, tests/test_member_coverage_gpu.py under repetition).
  // Here: everything but tile 3; from tile 2 on the loop's count holds (A4, A5, A6 + the chunk = 10 behind A3).
  pp_wait_vmcnt<HALF ? NPH : (D - 1) * NPH>();
  PP_BARRIER();
  read_words(0, 0);
  meta_read(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (zint) {
    meta_convert(ic<1>{}, s2c, zAc, zBc);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) decode(ic<1>{}, rawc[0][nf][0], s2c[nf], zAc[nf], zBc[nf], bw[0][nf]);
  } else {
    meta_convert(ic<0>{}, s2c, zAc, zBc);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) decode(ic<0>{}, rawc[0][nf][0], s2c[nf], zAc[nf], zBc[nf], bw[0][nf]);
  }
  if (grp == 1) PP_BARRIER();             // waves 4-7 run one segment behind waves 0-3

  // ---- main loop: four k-tiles (one weight chunk) per trip; every index below is a compile-time constant.
  // Phase p of a tile: MFMA jj = p >> 1 of the tile, activation fragments 8 * (p & 1) .. + 8, both weight fragments: 16 MFMAs.
  int slot = 0;                            // ring slot of the k-tile in hand
  auto load_segment = [&](auto TQ, auto PH, int t) {
    constexpr int tq = decltype(TQ)::value, p = decltype(PH)::value;
    constexpr int jj = p / HP, mh = p % HP;
    const unsigned char* sl = a_ring + slot * P::A_SLOT;
    if constexpr (tq == 0) stamp(t, p * 4 + 0);
    if constexpr (P::OPT & PPO_ABL_NOREAD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(afrag[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) afrag[(p & 1) * AF2 + i] = *reinterpret_cast<const u32x4*>(sl + a_rd[jj] + (mh * 8 + i) * (16 * P::TILE_ROW));
    }
    if constexpr (!(P::OPT & PPO_ABL_NODMA)) {
      const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
      dma_a(t + D, dslot, p);
      // the next chunk: its four pieces ride with tile 2 (this chunk's words are all in registers by then).  128-row tile: the
      // chunk after next, behind the last activation piece of the tile, into the buffer this chunk has left
      if constexpr (!HALF && tq == 2) dma_w((t >> 2) + 1, p, 0);
      if constexpr (HALF && tq == 2 && p == 1) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_w((t >> 2) + 2, pc, (t >> 2) & 1);
      }
    }
    if constexpr (tq == 0) stamp(t, p * 4 + 1);
    if constexpr (!HALF && p == 1) {
      // the lightest load segment picks up what compute segments 2 and 3 decode from: the next chunk's words (its pieces went
      // out with tile 2; private to the wave, so its own counter is all the ordering they need: only the two activation pieces
      // issued since may be outstanding) and, on odd tiles, the next body's Scale / Zeros (their window landed long ago)
      if constexpr (tq == 3) {
        if constexpr (!(P::OPT & PPO_ABL_NODMA)) pp_wait_vmcnt<2>();
        read_words(0, 0);
      }
      if constexpr (tq == 1) read_words(1, 0);      // second half of this chunk
      if constexpr ((tq & 1) == 1) meta_read((t + 1) >> 1);
    }
    if constexpr (!HALF && p == 2) {
      // everything of tile t + 1 (and older) has landed when at most the pieces issued in segments 0..2 of this tile (and, with
      // a ring of 4, in tile t - 1) are outstanding; a metadata window issued in between only makes the wait cover one operation more
      if constexpr (!(P::OPT & PPO_ABL_NODMA))
        pp_wait_vmcnt<(D - 2) * 4 + 3 + (tq == 2 ? 3 : 0) + ((D > 2 && tq == 3) ? 4 : 0)>();
    }
    if constexpr (HALF && p == 0) {
      // what compute segment 1 decodes from.  The chunk read here went out a trip and more ago: older than the activation
      // tile the previous k-tile's wait completed, so it has landed (vmcnt retires in order)
      if constexpr (tq == 3) read_words(0, ((t >> 2) + 1) & 1);
      if constexpr (tq == 1) read_words(1, (t >> 2) & 1);
      if constexpr ((tq & 1) == 1) meta_read((t + 1) >> 1);
    }
    if constexpr (HALF && p == 1) {
      // k-tile t + 1 went out in tile t - 3; since then: two activation pieces per tile of t - 2 .. t and the four weight pieces
      // of whichever of t - 3 .. t is a chunk's third tile (they follow that tile's last activation piece)
      static_assert(!HALF || D == 4, "the count below");
      if constexpr (!(P::OPT & PPO_ABL_NODMA)) pp_wait_vmcnt<10>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (tq == 0) stamp(t, p * 4 + 2);
    if constexpr (!((P::OPT & PPO_ABL_LONGSEG) && (p & 1) == 0)) PP_BARRIER();
  };
  auto compute_segment = [&](auto ZI, auto TQ, auto PH, int t) {
    constexpr int tq = decltype(TQ)::value, p = decltype(PH)::value;
    constexpr int jj = p / HP, mh = p % HP;
    constexpr int par = jj;                // operand pair in use; the other one is being decoded
    if constexpr (tq == 0) stamp(t, p * 4 + 3);
    // the next pair of operands, one weight fragment per phase: MFMA 1 of this tile, then MFMA 0 of the next tile (with the
    // next body's Scale / Zeros after an odd tile)
    constexpr int nf_dec = mh;
    if constexpr (HALF) {
      // one phase per MFMA of the tile: both weight fragments are decoded in it
      if constexpr (P::OPT & PPO_ABL_NODEC) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bw[par ^ 1][nf][i] = bw[par][nf][i];
            asm volatile("" : "+v"(bw[par ^ 1][nf][i]));
          }
      } else if constexpr (jj == 0) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) decode(ZI, rawc[tq][nf][1], s2c[nf], zAc[nf], zBc[nf], bw[1][nf]);
      } else {
        if constexpr ((tq & 1) == 1 && !(P::OPT & PPO_ABL_METAONCE)) meta_convert(ZI, s2c, zAc, zBc);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) decode(ZI, rawc[(tq + 1) & 3][nf][0], s2c[nf], zAc[nf], zBc[nf], bw[0][nf]);
      }
    } else if constexpr (P::OPT & PPO_ABL_NODEC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bw[par ^ 1][nf_dec][i] = bw[par][nf_dec][i];
        asm volatile("" : "+v"(bw[par ^ 1][nf_dec][i]));
      }
    } else if constexpr (jj == 0) {
      decode(ZI, rawc[tq][nf_dec][1], s2c[nf_dec], zAc[nf_dec], zBc[nf_dec], bw[1][nf_dec]);
    } else if constexpr ((tq & 1) == 0) {
      decode(ZI, rawc[tq + 1][nf_dec][0], s2c[nf_dec], zAc[nf_dec], zBc[nf_dec], bw[0][nf_dec]);
    } else {
      // after an odd tile the next tile opens a new k-body: segments 0 and 1 were the last to decode with the old values
      if constexpr (mh == 0 && !(P::OPT & PPO_ABL_METAONCE)) meta_convert(ZI, s2c, zAc, zBc);
      decode(ZI, rawc[(tq + 1) & 3][nf_dec][0], s2c[nf_dec], zAc[nf_dec], zBc[nf_dec], bw[0][nf_dec]);   // (tq == 3: the half chunk read in load segment 1)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const u32x4 bv = {bw[par][nf][0], bw[par][nf][1], bw[par][nf][2], bw[par][nf][3]};
        if constexpr (BF)
          acc[mh * 8 + i][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, afrag[(p & 1) * AF2 + i]),
                                                                      acc[mh * 8 + i][nf], 0, 0, 0);
        else if constexpr (F16)
          acc[mh * 8 + i][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv), __builtin_bit_cast(half8_t, afrag[(p & 1) * AF2 + i]),
                                                                     acc[mh * 8 + i][nf], 0, 0, 0);
        else
          acc[mh * 8 + i][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, bv), __builtin_bit_cast(i32x4, afrag[(p & 1) * AF2 + i]),
                                                                    acc[mh * 8 + i][nf], 0, 0, 0);
      }
    }
    // two MFMAs (32 matrix-pipe cycles), then up to three of the decode's vector operations in their shadow
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, HALF ? 5 : 3, 0);
    }
    if constexpr (!((P::OPT & PPO_ABL_LONGSEG) && (p & 1) == 0)) PP_BARRIER();
  };
  auto tile = [&](auto ZI, auto TQ, int t) {
    if constexpr ((P::OPT & PPO_ABL_LONGSEG) != 0 && NPH == 4) {
      load_segment(TQ, ic<0>{}, t);
      load_segment(TQ, ic<1>{}, t);
      compute_segment(ZI, TQ, ic<0>{}, t);
      compute_segment(ZI, TQ, ic<1>{}, t);
      load_segment(TQ, ic<2>{}, t);
      load_segment(TQ, ic<3>{}, t);
      compute_segment(ZI, TQ, ic<2>{}, t);
      compute_segment(ZI, TQ, ic<3>{}, t);
      slot = slot + 1 == RING ? 0 : slot + 1;
      return;
    }
    load_segment(TQ, ic<0>{}, t);
    compute_segment(ZI, TQ, ic<0>{}, t);
    load_segment(TQ, ic<1>{}, t);
    compute_segment(ZI, TQ, ic<1>{}, t);
    if constexpr (NPH == 4) {
      load_segment(TQ, ic<(NPH == 4 ? 2 : 0)>{}, t);
      compute_segment(ZI, TQ, ic<(NPH == 4 ? 2 : 0)>{}, t);
      load_segment(TQ, ic<(NPH == 4 ? 3 : 0)>{}, t);
      compute_segment(ZI, TQ, ic<(NPH == 4 ? 3 : 0)>{}, t);
    }
    slot = slot + 1 == RING ? 0 : slot + 1;
  };
  auto main_loop = [&](auto ZI) {
    if constexpr (P::OPT & PPO_ABL_NOREAD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) afrag[i] = *reinterpret_cast<const u32x4*>(a_ring + a_rd[0] + i * (16 * P::TILE_ROW));
    }
    // A metadata window (8 groups) lasts 16 << gq_shift k-tiles; the next one is asked for a whole window ahead, between two
    // trips (the trip itself stays free of branches: one basic block per segment)
    const int wtiles = 16 << a.gq_shift;
    for (int t = 0; t < ntiles; t += 4) {
      if constexpr (P::HAS_META) {
        if ((t & (wtiles - 1)) == 0) dma_meta((t >> (4 + a.gq_shift)) + 1);
      }
      tile(ZI, ic<0>{}, t);
      tile(ZI, ic<1>{}, t + 1);
      tile(ZI, ic<2>{}, t + 2);
      tile(ZI, ic<3>{}, t + 3);
    }
  };
  if constexpr ((P::OPT & PPO_PRIO) != 0) {
    if (grp == 1) __builtin_amdgcn_s_setprio(1);
  }
  if (zint) main_loop(ic<1>{});
  else main_loop(ic<0>{});
  if constexpr ((P::OPT & PPO_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
  if (grp == 0) PP_BARRIER();
  if constexpr (P::OPT & PPO_TRACE) {
    if (lane == 0 && a.lut) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<void*>(a.lut)) + ((long)blockIdx.x * 8 + wave) * 20;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i] = tr[i];
      dst[16] = tr_c0;
      dst[17] = tr_r0;
      dst[18] = __builtin_readcyclecounter();
      dst[19] = __builtin_amdgcn_s_memrealtime();
    }
  }

  // ---- epilogue: the 256 x 256 tile leaves through LDS, whole rows per store ----
  // accumulator (f, nf): activation row m = 16 f + fr, weight rows n = 32 wave + 16 nf + 4 kb + {0..3}
  pp_wait_vmcnt<0>();                      // the clamped tail pieces still write ring slots
  PP_BARRIER();
  const int el = pp_opaque(lane);          // (lane roles recomputed: nothing of the epilogue stays live across the loop)
  const int e_fr = el & 15, e_kb = el >> 4, e_ln = el & 31, e_h = el >> 5;
  // 4-byte output elements: float32 out of the float members; int32 out of the integer ones - unless the caller's epilogue
  // (wqaa_matmul_ex: out / row_scale[m] / tensor_scale -> float16, integration/BitNet/utils_quant.py:205-216) turns the int32
  // sums into float16 on their way out: then the tile leaves as the float members' does, half the bytes and one pass
  const bool wide_out = F16 ? a.out_dtype == WQAA_F32 : a.epi_row == nullptr;
  if (!wide_out) {
   {
    // unit = 4 consecutive n (8 bytes); unit u of row m lives at pair ((u >> 1) ^ (m & 7)), half ((u & 1) ^ ((m >> 3) & 1)):
    // the 16 lanes of a ds_write_b64 group (16 consecutive m, one u) hit 16 distinct 8-byte slots of a 128-byte window
    half_t bias_h[2][4];
    float bias_bf[2][4];                   // bfloat16: cast, then + bias in bfloat16 (store_out, wqaa_kinds.h)
    if (a.has_bias) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = nw0 + nf * 16 + e_kb * 4 + i;
          if constexpr (BF) bias_bf[nf][i] = bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[n < a.N ? n : a.N - 1]);
          else bias_h[nf][i] = reinterpret_cast<const half_t*>(a.bias)[n < a.N ? n : a.N - 1];
        }
    }
    // integer members: out / row_scale[m] / tensor_scale by the shared-divisor form of the IEEE division (ExactDiv, wqaa_kinds.h)
    // when every divisor this wave meets is in its range (always, for scales that come from a quantiser), `a / b` otherwise
    bool div_safe = false;
    ExactDiv dts{1.f, 1.f};
    if constexpr (!F16) {
      bool ok = ExactDiv::safe(a.epi_tensor);
#pragma unroll
      for (int f = 0; f < NMF; ++f) ok = ok && ExactDiv::safe(a.epi_row[m0 + f * 16 + e_fr < a.M ? m0 + f * 16 + e_fr : a.M - 1]);
      div_safe = __all(ok);
      dts = ExactDiv::prepare(a.epi_tensor);
    }
#pragma unroll
    for (int f = 0; f < NMF; ++f) {
      const int m = f * 16 + e_fr;
      float rs = 1.f;                      // (integer members: the row's activation scale, one load per 8 outputs)
      if constexpr (!F16) rs = a.epi_row[m0 + m < a.M ? m0 + m : a.M - 1];
      const ExactDiv drs = ExactDiv::prepare(rs);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        uint32_t lo_u, hi_u;
        if constexpr (!F16) {
          // store_out_fused (wqaa_kinds.h) to the letter: two IEEE fp32 divisions, cast, + bias in float16
          half_t v[4];
          if (div_safe) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const ExactDiv::f32x2 x = dts(drs(ExactDiv::f32x2{(float)acc[f][nf][i], (float)acc[f][nf][i + 1]}));
              v[i] = (half_t)x[0];
              v[i + 1] = (half_t)x[1];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (half_t)(((float)acc[f][nf][i] / rs) / a.epi_tensor);
          }
          if (a.has_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] + bias_h[nf][i];
          }
          const half2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
          lo_u = as_u32(lo);
          hi_u = as_u32(hi);
        } else if constexpr (BF) {
          float x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[i] = bf16_round(acc[f][nf][i]);
            if (a.has_bias) x[i] = bf16_round(x[i] + bias_bf[nf][i]);
          }
          lo_u = (__builtin_bit_cast(uint32_t, x[0]) >> 16) | (__builtin_bit_cast(uint32_t, x[1]) & 0xFFFF0000u);
          hi_u = (__builtin_bit_cast(uint32_t, x[2]) >> 16) | (__builtin_bit_cast(uint32_t, x[3]) & 0xFFFF0000u);
        } else {
          half_t v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = (half_t)acc[f][nf][i];
            if (a.has_bias) v[i] = v[i] + bias_h[nf][i];
          }
          const half2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
          lo_u = as_u32(lo);
          hi_u = as_u32(hi);
        }
        const int u = wave * 8 + nf * 4 + e_kb;
        const int up = (((u >> 1) ^ (m & 7)) << 1) | ((u & 1) ^ ((m >> 3) & 1));
        *reinterpret_cast<u32x2*>(smem + m * 512 + up * 8) = u32x2{lo_u, hi_u};
      }
    }
    PP_FENCE();
    __syncthreads();
    // a wave stores its 32 (16) rows, two per instruction: lane c = lane & 31 takes the 16-byte pair c of row m
#pragma unroll
    for (int rr = 0; rr < WROWS / 2; ++rr) {
      const int m = wave * WROWS + rr * 2 + e_h;
      const int c = e_ln;
      u32x4 x = *reinterpret_cast<const u32x4*>(smem + m * 512 + ((c ^ (m & 7)) * 16));
      if ((m >> 3) & 1) x = u32x4{x[2], x[3], x[0], x[1]};
      const int n = n0 + c * 8;
      if (m0 + m < a.M && n < a.N) {
        u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<half_t*>(a.C) + (long)(m0 + m) * a.N + n);
        pp_store_out(dst, x, a.ws_policy);
      }
    }
   }
  } else {
    // int32 / float32 output: 16 bytes per (lane, fragment); passes of 128 rows (128 KiB each).  Slot s = n / 4 of row m lives at
    // s ^ (m & 7): the 8 lanes of a ds_write_b128 group (8 consecutive m) hit 8 distinct 16-byte slots
    using bias_t = typename std::conditional<F16, float, int>::type;
    bias_t bias_i[2][4];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nw0 + nf * 16 + e_kb * 4 + i;
        const int nc = n < a.N ? n : a.N - 1;
        if constexpr (BF) bias_i[nf][i] = a.has_bias ? bf16_bits_to_float(reinterpret_cast<const uint16_t*>(a.bias)[nc]) : 0.f;
        else if constexpr (F16) bias_i[nf][i] = a.has_bias ? (float)reinterpret_cast<const half_t*>(a.bias)[nc] : 0.f;
        else bias_i[nf][i] = a.has_bias ? (int)reinterpret_cast<const int8_t*>(a.bias)[nc] : 0;
      }
#pragma unroll
    for (int pass = 0; pass < P::BM / 128; ++pass) {
      if (pass) __syncthreads();
#pragma unroll
      for (int ff = 0; ff < 8; ++ff) {
        const int f = pass * 8 + ff;
        const int ml = ff * 16 + e_fr;       // row inside the pass
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const int sidx = wave * 8 + nf * 4 + e_kb;
          const acc_t v = {acc[f][nf][0] + bias_i[nf][0], acc[f][nf][1] + bias_i[nf][1], acc[f][nf][2] + bias_i[nf][2], acc[f][nf][3] + bias_i[nf][3]};
          *reinterpret_cast<acc_t*>(smem + ml * 1024 + ((sidx ^ (ml & 7)) * 16)) = v;
        }
      }
      PP_FENCE();
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ml = wave * 16 + rr;
        const i32x4 x = *reinterpret_cast<const i32x4*>(smem + ml * 1024 + ((el ^ (ml & 7)) * 16));
        const int m = m0 + pass * 128 + ml, n = n0 + el * 4;
        if (m < a.M && n < a.N) pp_store_out(reinterpret_cast<i32x4*>(reinterpret_cast<int*>(a.C) + (long)m * a.N + n), x, a.ws_policy);
      }
    }
  }
#endif
}


// ------------------------------------------------------------------------------------------
// dense fp8 x fp8 (BASELINE c5; replaces tilelang/dense/matmul_mma.py:220-320 for M >= 256): the same skeleton with
// nothing to decode.  v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (the 128-deep form runs at twice the 32-deep
// form's rate, and the 16x16 shape sustains ~6 % more than 32x32 under the power limit: tools/mfma_power.hip): one k-tile
// (128 k = one 128-byte line of every row of BOTH operands) is one instruction per fragment pair.  A lane feeds k-slots
// [16 kb, +16) and [64 + 16 kb, +16) of the tile: granules kb and 4 + kb, whose two 8-lane halves of a ds_read_b128 service
// group never meet on a 16-byte slot.  A wave keeps its own 32 weight rows in a private two-slot LDS ring (4 KiB per k-tile,
// LDS-DMA in full lines, swizzled like the activations) and reads them into registers at the head of the tile.
// Phase p of a tile: activation fragments 4p .. 4p+3 x both weight fragments = 8 MFMAs of 32 matrix-pipe cycles.
// ------------------------------------------------------------------------------------------
// BM_ = 128: the same loop on half the activation rows (two phases per k-tile) for shapes whose 256-row tiles leave the chip
// short of whole rounds - the N / 8 column shards of BASELINE c5 first of all (4096 x 1024 x 8192 is 64 tiles of 256 x 256).
// A k-tile then lasts half as long: activation ring of 4 (16 KiB slots), weight ring of 3 per wave, asked for three tiles ahead.
// WFMT_ / AFMT_ 2, 3 (round 4): dense float16 / bfloat16 x the same type - W "stored in A_dtype" (the reference's plain matmul,
// tilelang/dense/matmul_mma.py, and the second pass of the two-pass member: B_decode, then this).  Same skeleton, same bytes per
// k-tile (64 k = one 128-byte line of every row of both operands); a fragment pair is two v_mfma_f32_16x16x32_f16 / _bf16
// (granule kb = k [8 kb, +8) of the tile's first 32, granule 4 + kb the same of its second 32: ascending k, the order of the
// fused members).
template <int WFMT_, int AFMT_, int OPT_ = 0, int BM_ = 256>
struct PP8Policy {
  static constexpr int WFMT = WFMT_, AFMT = AFMT_, OPT = OPT_;   // 0 e4m3, 1 e5m2; 2 float16, 3 bfloat16, 4 int8 (both operands)
  static constexpr int ESZ = (WFMT_ == 2 || WFMT_ == 3) ? 2 : 1; // bytes per element
  static constexpr bool I8 = WFMT_ == 4;                         // int8 x int8 -> int32 (the reference's INT8 x INT8 row; two
                                                                 // v_mfma_i32_16x16x64_i8 per fragment pair; int32 output in 128-row passes)
  static_assert(WFMT_ < 2 || WFMT_ == AFMT_, "the 16-bit / int8 members take one type for both operands");
  static constexpr int BM = BM_, BN = 256, THREADS = 512, KT = 128 / ESZ, TILE_ROW = 128;
  static constexpr int RING = BM_ == 128 ? 4 : 3, D = RING - 1;
  static constexpr int WS = BM_ == 128 ? 3 : 2;      // weight k-tile slots per wave (4 KiB each)
  static constexpr int NPH = BM_ / 64;               // phases (8 MFMAs per wave each) per k-tile
  static constexpr int A_SLOT = BM * TILE_ROW;
  static constexpr int W_OFF = RING * A_SLOT;
  static constexpr int LDS_BYTES = W_OFF + 8 * WS * 4096;
  static_assert(BM_ == 256 || BM_ == 128, "256- or 128-row tile");
  static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= BM * BN * 2, "LDS budget / output staging");
};

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_pp8_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = P::D, RING = P::RING, NPH = P::NPH, WS = P::WS;
  constexpr bool HALF = P::BM == 128;
  constexpr int NMF = P::BM / 16, WROWS = P::BM / 8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int fr = lane & 15, kb = lane >> 4;

  const TileOfBlock tob = tile_of_block(a, (int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tob.tile_m, tile_n = tob.tile_n;
  const int m0 = tile_m * P::BM, n0 = (tile_n + a.tile_n_off) * P::BN, nw0 = n0 + wave * 32;
  const int ntiles = a.K / P::KT;
  const uint32_t rowb = (uint32_t)a.K * (uint32_t)P::ESZ;     // bytes per row of either operand

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((long)a.M * rowb), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.B), 0, (int)((long)a.N * rowb), 0x00020000);
  // piece j of either operand: 8 rows x one 128-byte line; rows beyond the matrix are out of the buffer's range (read as zero)
  const int g0_ = (lane & 7) ^ ((lane >> 4) & 7);
  const uint32_t v0 = (uint32_t)(wave * 32 + (lane >> 3)) * rowb + (uint32_t)(g0_ * 16);         // the wave's weight rows
  const uint32_t v0a = (uint32_t)(wave * WROWS + (lane >> 3)) * rowb + (uint32_t)(g0_ * 16);     // ... and its share of the activation rows
  const int vd = ((g0_ ^ 4) - g0_) * 16;
  const uint32_t a_rows0 = (uint32_t)m0 * rowb, w_rows0 = (uint32_t)n0 * rowb;

  unsigned char* const a_ring = smem;
  unsigned char* const w_ring = smem + P::W_OFF + wave * (WS * 4096);
  auto dma_a = [&](int tt, int slot, int j) {
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = a_ring + slot * P::A_SLOT + (wave * WROWS + j * 8) * P::TILE_ROW;
    const uint32_t rows = a_rows0 + (uint32_t)(j * 8) * rowb;
    const uint32_t voff = (j & 1) ? v0a + (uint32_t)vd + rows : v0a + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };
  auto dma_w = [&](int tt, int j, int ws) {          // piece j of the wave's 32 weight rows of k-tile tt -> weight slot ws
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = w_ring + ws * 4096 + j * 1024;
    const uint32_t rows = w_rows0 + (uint32_t)(j * 8) * rowb;
    const uint32_t voff = (j & 1) ? v0 + (uint32_t)vd + rows : v0 + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };

  const int swl = (fr >> 1) & 7;
  uint32_t rd[2];                                    // granules kb and 4 + kb of row fr
  rd[0] = (uint32_t)(fr * P::TILE_ROW + ((kb ^ swl) * 16));
  rd[1] = (uint32_t)(fr * P::TILE_ROW + (((4 + kb) ^ swl) * 16));

  using acc_t = typename std::conditional<P::I8, i32x4, f32x4>::type;
  acc_t acc[NMF][2];
#pragma unroll
  for (int f = 0; f < NMF; ++f)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) acc[f][nf] = acc_t{0, 0, 0, 0};
  u32x4 afrag[4][2], wfrag[2][2];

  // ---- prologue ----
  if constexpr (HALF) {
    // three k-tiles in flight, each as [activation piece 0, piece 1, four weight pieces]: the order the loop keeps
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      dma_a(tt, tt, 0);
      dma_a(tt, tt, 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_w(tt, j, tt);
    }
    pp_wait_vmcnt<12>();
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w(0, j, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(0, 0, j);
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_w(1, j, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(1, 1, j);
    pp_wait_vmcnt<7>();
  }
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();

  int slot = 0;
  int wslot = 0;                                     // weight slot of the k-tile in hand (128-row tile: t % 3; 256-row: t & 1)
  auto load_segment = [&](auto PH, int t) {
    constexpr int p = decltype(PH)::value;
    const unsigned char* sl = a_ring + slot * P::A_SLOT;
    if constexpr (p == 0) {                          // this tile's weights (landed: the wait of the previous tile's last segment)
      const unsigned char* ws = w_ring + wslot * 4096;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int i = 0; i < 2; ++i) wfrag[nf][i] = *reinterpret_cast<const u32x4*>(ws + nf * 2048 + rd[i]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i) afrag[f][i] = *reinterpret_cast<const u32x4*>(sl + (p * 4 + f) * (16 * P::TILE_ROW) + rd[i]);
    if constexpr (HALF) {
      // the k-tile three ahead, as [activation piece 0] [activation piece 1, four weight pieces]: its weights take the slot of the
      // tile in hand, whose fragments went to registers in segment 0
      const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
      dma_a(t + D, dslot, p);
      if constexpr (p == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_w(t + 3, j, wslot);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // tile t + 1 complete when only the two younger tiles (six operations each) are outstanding
      if constexpr (p == 1) pp_wait_vmcnt<12>();
    } else {
      // weights: piece 3 of tile t + 1 rides with segment 0 (its slot is the one tile t - 1 has left), pieces 0..2 of tile t + 2
      // with segments 1..3 (this tile's weights are in registers by then)
      if constexpr (p == 0) dma_w(t + 1, 3, (t + 1) & 1);
      else dma_w(t + 2, p - 1, t & 1);
      {
        const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
        dma_a(t + D, dslot, p);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // tile t + 1 complete (activations and weights) when only what followed its last weight piece is outstanding
      if constexpr (p == 2) pp_wait_vmcnt<5>();
    }
    PP_BARRIER();
  };
  auto compute_segment = [&](auto PH, int t) {
    constexpr int p = decltype(PH)::value;
    if constexpr (P::I8) {
      // int8 operands: the tile's two 64-deep halves, in ascending k
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wfrag[nf][i]), __builtin_bit_cast(i32x4, afrag[f][i]),
                                                                       acc[p * 4 + f][nf], 0, 0, 0);
    } else if constexpr (P::ESZ == 2) {
      // 16-bit operands: the tile's two 32-deep halves, in ascending k
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (P::WFMT == 3)
              acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[nf][i]), __builtin_bit_cast(bf16x8_t, afrag[f][i]),
                                                                          acc[p * 4 + f][nf], 0, 0, 0);
            else
              acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wfrag[nf][i]), __builtin_bit_cast(half8_t, afrag[f][i]),
                                                                         acc[p * 4 + f][nf], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const i32x8 av = {(int)afrag[f][0][0], (int)afrag[f][0][1], (int)afrag[f][0][2], (int)afrag[f][0][3],
                          (int)afrag[f][1][0], (int)afrag[f][1][1], (int)afrag[f][1][2], (int)afrag[f][1][3]};
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const i32x8 wv = {(int)wfrag[nf][0][0], (int)wfrag[nf][0][1], (int)wfrag[nf][0][2], (int)wfrag[nf][0][3],
                            (int)wfrag[nf][1][0], (int)wfrag[nf][1][1], (int)wfrag[nf][1][2], (int)wfrag[nf][1][3]};
          acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, acc[p * 4 + f][nf], P::WFMT, P::AFMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
      }
    }
    PP_BARRIER();
  };
  for (int t = 0; t < ntiles; ++t) {
    load_segment(ic<0>{}, t);
    compute_segment(ic<0>{}, t);
    load_segment(ic<1>{}, t);
    compute_segment(ic<1>{}, t);
    if constexpr (NPH == 4) {
      load_segment(ic<(NPH == 4 ? 2 : 0)>{}, t);
      compute_segment(ic<(NPH == 4 ? 2 : 0)>{}, t);
      load_segment(ic<(NPH == 4 ? 3 : 0)>{}, t);
      compute_segment(ic<(NPH == 4 ? 3 : 0)>{}, t);
    }
    slot = slot + 1 == RING ? 0 : slot + 1;
    wslot = wslot + 1 == WS ? 0 : wslot + 1;
  }
  if (grp == 0) PP_BARRIER();

  // ---- epilogue: as wq_gemm_pp_kernel (float16 out of fp32 accumulators; the reference defines no fp8 bias) ----
  pp_wait_vmcnt<0>();
  PP_BARRIER();
  const int el = pp_opaque(lane);
  const int e_fr = el & 15, e_kb = el >> 4, e_ln = el & 31, e_h = el >> 5;
  if constexpr (P::I8) {
    // int32 output (+ int8 bias): wq_gemm_pp_kernel's wide pass - 16 bytes per (lane, fragment), passes of 128 rows; slot s = n / 4
    // of row m lives at s ^ (m & 7)
    int bias_i[2][4];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nw0 + nf * 16 + e_kb * 4 + i;
        bias_i[nf][i] = a.has_bias ? (int)reinterpret_cast<const int8_t*>(a.bias)[n < a.N ? n : a.N - 1] : 0;
      }
#pragma unroll
    for (int pass = 0; pass < P::BM / 128; ++pass) {
      if (pass) __syncthreads();
#pragma unroll
      for (int ff = 0; ff < 8; ++ff) {
        const int f = pass * 8 + ff;
        const int ml = ff * 16 + e_fr;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const int sidx = wave * 8 + nf * 4 + e_kb;
          const i32x4 v = {(int)acc[f][nf][0] + bias_i[nf][0], (int)acc[f][nf][1] + bias_i[nf][1], (int)acc[f][nf][2] + bias_i[nf][2],
                           (int)acc[f][nf][3] + bias_i[nf][3]};
          *reinterpret_cast<i32x4*>(smem + ml * 1024 + ((sidx ^ (ml & 7)) * 16)) = v;
        }
      }
      PP_FENCE();
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ml = wave * 16 + rr;
        const i32x4 x = *reinterpret_cast<const i32x4*>(smem + ml * 1024 + ((el ^ (ml & 7)) * 16));
        const int m = m0 + pass * 128 + ml, n = n0 + el * 4;
        if (m < a.M && n < a.N) pp_store_out(reinterpret_cast<i32x4*>(reinterpret_cast<int*>(a.C) + (long)m * a.N + n), x, a.ws_policy);
      }
    }
    return;
  }
#pragma unroll
  for (int f = 0; f < NMF; ++f) {
    const int m = f * 16 + e_fr;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      uint32_t lo_u, hi_u;
      if constexpr (P::WFMT == 3) {         // bfloat16 out: round to nearest even (bf16_round), packed pairs
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = bf16_round(acc[f][nf][i]);
        lo_u = (__builtin_bit_cast(uint32_t, x[0]) >> 16) | (__builtin_bit_cast(uint32_t, x[1]) & 0xFFFF0000u);
        hi_u = (__builtin_bit_cast(uint32_t, x[2]) >> 16) | (__builtin_bit_cast(uint32_t, x[3]) & 0xFFFF0000u);
      } else {
        const half2_t lo = {(half_t)acc[f][nf][0], (half_t)acc[f][nf][1]}, hi = {(half_t)acc[f][nf][2], (half_t)acc[f][nf][3]};
        lo_u = as_u32(lo);
        hi_u = as_u32(hi);
      }
      const int u = wave * 8 + nf * 4 + e_kb;
      const int up = (((u >> 1) ^ (m & 7)) << 1) | ((u & 1) ^ ((m >> 3) & 1));
      *reinterpret_cast<u32x2*>(smem + m * 512 + up * 8) = u32x2{lo_u, hi_u};
    }
  }
  PP_FENCE();
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < WROWS / 2; ++rr) {
    const int m = wave * WROWS + rr * 2 + e_h;
    u32x4 x = *reinterpret_cast<const u32x4*>(smem + m * 512 + ((e_ln ^ (m & 7)) * 16));
    if ((m >> 3) & 1) x = u32x4{x[2], x[3], x[0], x[1]};
    const int n = n0 + e_ln * 8;
    if (m0 + m < a.M && n < a.N) pp_store_out(reinterpret_cast<u32x4*>(reinterpret_cast<half_t*>(a.C) + (long)(m0 + m) * a.N + n), x, a.ws_policy);
  }
#endif
}

// ------------------------------------------------------------------------------------------
// Round 5 (VERDICT r04 "next" #3a): the dense 256 x 256 tile on a 2 (m) x 4 (n) WAVE GRID - a wave owns 64 weight rows (four
// fragments) x 128 activation rows (eight fragments).  wq_gemm_pp8_kernel's 1 x 8 grid reads every activation fragment in every
// wave: per k-tile and wave 32 activation + 4 weight reads of 16 bytes for its 64 fragment pairs; here 16 + 8 - a third fewer LDS
// reads per MFMA, which on a power-limited chip is time (tools/mfma_power.hip: one ds_read_b128 per 32 matrix cycles costs 17-19 %).
// The price: both operands are SHARED tiles now (the weight rows of a wave pair w, w + 4 are the same 64), so the weights get the
// activations' discipline - a ring of whole 256-row tiles in LDS (2 slots of 32 KiB next to the activations' 3: exactly the CU's
// 160 KiB), every wave copies its 32 rows of each, waits for its own pieces, and a barrier stands between "landed" and "read".
// Slot re-use across the two role groups (one barrier apart): W(t + 2) goes into W(t)'s slot from load segment 1 of tile t on -
// two barriers after the issuing group's own read of W(t), one after the other group's; A(t + 2) into A(t - 1)'s slot from load
// segment 0 of tile t on - likewise behind both groups' last reads of it.  Per k-tile and wave 8 LDS-DMA pieces in one order
// ([A p0, p1] with segment 0, [A p2, p3, W p0..p3] with segment 1), so "tile t + 1 has landed" is `vmcnt(8)` at the end of segment 1.
// Two phases per k-tile (4 activation fragments x 4 weight fragments each: 16 pairs = 16 / 32 MFMAs).  Same k order per output as
// the 1 x 8 grid (ascending k inside a tile, tiles in order): bit-identical results.
// ------------------------------------------------------------------------------------------
template <int WFMT_, int AFMT_>
struct PP8WPolicy {
  static constexpr int WFMT = WFMT_, AFMT = AFMT_;
  static constexpr int ESZ = (WFMT_ == 2 || WFMT_ == 3) ? 2 : 1;
  static constexpr bool I8 = WFMT_ == 4;
  static_assert(WFMT_ < 2 || WFMT_ == AFMT_, "the 16-bit / int8 members take one type for both operands");
  static constexpr int BM = 256, BN = 256, THREADS = 512, KT = 128 / ESZ, TILE_ROW = 128;
  static constexpr int RING = 3, D = 2, WS = 2;
  static constexpr int A_SLOT = BM * TILE_ROW, W_SLOT = BN * TILE_ROW;
  static constexpr int W_OFF = RING * A_SLOT;
  static constexpr int LDS_BYTES = W_OFF + WS * W_SLOT;
  static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= BM * BN * 2, "LDS budget / output staging");
};

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_pp8w_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = P::D, RING = P::RING;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                         // role group = M-half: waves w and w + 4 share a SIMD and a set of weight rows
  const int nq = wave & 3;
  const int fr = lane & 15, kb = lane >> 4;

  const TileOfBlock tob = tile_of_block(a, (int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tob.tile_m, tile_n = tob.tile_n;
  const int m0 = tile_m * P::BM, n0 = (tile_n + a.tile_n_off) * P::BN;
  const int ntiles = a.K / P::KT;
  const uint32_t rowb = (uint32_t)a.K * (uint32_t)P::ESZ;

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((long)a.M * rowb), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.B), 0, (int)((long)a.N * rowb), 0x00020000);
  // piece j of either operand: 8 rows x one 128-byte line of this wave's 32-row share of the tile; rows beyond the matrix are out of
  // the buffer's range (read as zero)
  const int g0_ = (lane & 7) ^ ((lane >> 4) & 7);
  const uint32_t v0 = (uint32_t)(wave * 32 + (lane >> 3)) * rowb + (uint32_t)(g0_ * 16);
  const int vd = ((g0_ ^ 4) - g0_) * 16;
  const uint32_t a_rows0 = (uint32_t)m0 * rowb, w_rows0 = (uint32_t)n0 * rowb;

  unsigned char* const a_ring = smem;
  unsigned char* const w_ring = smem + P::W_OFF;
  auto dma_a = [&](int tt, int slot, int j) {
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = a_ring + slot * P::A_SLOT + (wave * 32 + j * 8) * P::TILE_ROW;
    const uint32_t rows = a_rows0 + (uint32_t)(j * 8) * rowb;
    const uint32_t voff = (j & 1) ? v0 + (uint32_t)vd + rows : v0 + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };
  auto dma_w = [&](int tt, int ws, int j) {
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = w_ring + ws * P::W_SLOT + (wave * 32 + j * 8) * P::TILE_ROW;
    const uint32_t rows = w_rows0 + (uint32_t)(j * 8) * rowb;
    const uint32_t voff = (j & 1) ? v0 + (uint32_t)vd + rows : v0 + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };

  const int swl = (fr >> 1) & 7;
  uint32_t rd[2];                                    // granules kb and 4 + kb of row fr
  rd[0] = (uint32_t)(fr * P::TILE_ROW + ((kb ^ swl) * 16));
  rd[1] = (uint32_t)(fr * P::TILE_ROW + (((4 + kb) ^ swl) * 16));

  using acc_t = typename std::conditional<P::I8, i32x4, f32x4>::type;
  acc_t acc[8][4];                                   // [activation fragment of the wave's M-half][weight fragment of its 64 rows]
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[f][nf] = acc_t{0, 0, 0, 0};
  u32x4 afrag[4][2], wfrag[4][2];

  // ---- prologue: tiles 0 and 1, each as [A p0..p3, W p0..p3] ----
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(tt, tt, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w(tt, tt, j);
  }
  pp_wait_vmcnt<8>();
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();

  int slot = 0;
  auto load_segment = [&](auto PH, int t) {
    constexpr int p = decltype(PH)::value;
    const unsigned char* sl = a_ring + slot * P::A_SLOT + (grp * 8 + p * 4) * (16 * P::TILE_ROW);
    if constexpr (p == 0) {
      const unsigned char* ws = w_ring + (t & 1) * P::W_SLOT + nq * (64 * P::TILE_ROW);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int i = 0; i < 2; ++i) wfrag[nf][i] = *reinterpret_cast<const u32x4*>(ws + nf * (16 * P::TILE_ROW) + rd[i]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < 2; ++i) afrag[f][i] = *reinterpret_cast<const u32x4*>(sl + f * (16 * P::TILE_ROW) + rd[i]);
    const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
    if constexpr (p == 0) {
      dma_a(t + 2, dslot, 0);
      dma_a(t + 2, dslot, 1);
    } else {
      dma_a(t + 2, dslot, 2);
      dma_a(t + 2, dslot, 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_w(t + 2, t & 1, j);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (p == 1) pp_wait_vmcnt<8>();        // tile t + 1 complete: only tile t + 2's eight pieces may be outstanding
    PP_BARRIER();
  };
  auto compute_segment = [&](auto PH) {
    constexpr int p = decltype(PH)::value;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if constexpr (P::I8) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wfrag[nf][i]), __builtin_bit_cast(i32x4, afrag[f][i]),
                                                                       acc[p * 4 + f][nf], 0, 0, 0);
      } else if constexpr (P::ESZ == 2) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (P::WFMT == 3)
              acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[nf][i]), __builtin_bit_cast(bf16x8_t, afrag[f][i]),
                                                                          acc[p * 4 + f][nf], 0, 0, 0);
            else
              acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wfrag[nf][i]), __builtin_bit_cast(half8_t, afrag[f][i]),
                                                                         acc[p * 4 + f][nf], 0, 0, 0);
          }
      } else {
        const i32x8 av = {(int)afrag[f][0][0], (int)afrag[f][0][1], (int)afrag[f][0][2], (int)afrag[f][0][3],
                          (int)afrag[f][1][0], (int)afrag[f][1][1], (int)afrag[f][1][2], (int)afrag[f][1][3]};
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const i32x8 wv = {(int)wfrag[nf][0][0], (int)wfrag[nf][0][1], (int)wfrag[nf][0][2], (int)wfrag[nf][0][3],
                            (int)wfrag[nf][1][0], (int)wfrag[nf][1][1], (int)wfrag[nf][1][2], (int)wfrag[nf][1][3]};
          acc[p * 4 + f][nf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, acc[p * 4 + f][nf], P::WFMT, P::AFMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
      }
    }
    PP_BARRIER();
  };
  for (int t = 0; t < ntiles; ++t) {
    load_segment(ic<0>{}, t);
    compute_segment(ic<0>{});
    load_segment(ic<1>{}, t);
    compute_segment(ic<1>{});
    slot = slot + 1 == RING ? 0 : slot + 1;
  }
  if (grp == 0) PP_BARRIER();

  // ---- epilogue: wq_gemm_pp8_kernel's, with this grid's (row, column) of a fragment ----
  pp_wait_vmcnt<0>();
  PP_BARRIER();
  const int el = pp_opaque(lane);
  const int e_fr = el & 15, e_kb = el >> 4, e_ln = el & 31, e_h = el >> 5;
  if constexpr (P::I8) {
    int bias_i[4][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + nq * 64 + nf * 16 + e_kb * 4 + i;
        bias_i[nf][i] = a.has_bias ? (int)reinterpret_cast<const int8_t*>(a.bias)[n < a.N ? n : a.N - 1] : 0;
      }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {            // pass = M-half: the half's four waves stage its 128 rows, all eight store them
      if (pass) __syncthreads();
      if (grp == pass) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int ml = f * 16 + e_fr;
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) {
            const int sidx = nq * 16 + nf * 4 + e_kb;
            const i32x4 v = {(int)acc[f][nf][0] + bias_i[nf][0], (int)acc[f][nf][1] + bias_i[nf][1], (int)acc[f][nf][2] + bias_i[nf][2],
                             (int)acc[f][nf][3] + bias_i[nf][3]};
            *reinterpret_cast<i32x4*>(smem + ml * 1024 + ((sidx ^ (ml & 7)) * 16)) = v;
          }
        }
      }
      PP_FENCE();
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int ml = wave * 16 + rr;
        const i32x4 x = *reinterpret_cast<const i32x4*>(smem + ml * 1024 + ((el ^ (ml & 7)) * 16));
        const int m = m0 + pass * 128 + ml, n = n0 + el * 4;
        if (m < a.M && n < a.N) pp_store_out(reinterpret_cast<i32x4*>(reinterpret_cast<int*>(a.C) + (long)m * a.N + n), x, a.ws_policy);
      }
    }
    return;
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int m = (grp * 8 + f) * 16 + e_fr;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      uint32_t lo_u, hi_u;
      if constexpr (P::WFMT == 3) {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = bf16_round(acc[f][nf][i]);
        lo_u = (__builtin_bit_cast(uint32_t, x[0]) >> 16) | (__builtin_bit_cast(uint32_t, x[1]) & 0xFFFF0000u);
        hi_u = (__builtin_bit_cast(uint32_t, x[2]) >> 16) | (__builtin_bit_cast(uint32_t, x[3]) & 0xFFFF0000u);
      } else {
        const half2_t lo = {(half_t)acc[f][nf][0], (half_t)acc[f][nf][1]}, hi = {(half_t)acc[f][nf][2], (half_t)acc[f][nf][3]};
        lo_u = as_u32(lo);
        hi_u = as_u32(hi);
      }
      const int u = nq * 16 + nf * 4 + e_kb;
      const int up = (((u >> 1) ^ (m & 7)) << 1) | ((u & 1) ^ ((m >> 3) & 1));
      *reinterpret_cast<u32x2*>(smem + m * 512 + up * 8) = u32x2{lo_u, hi_u};
    }
  }
  PP_FENCE();
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int m = wave * 32 + rr * 2 + e_h;
    u32x4 x = *reinterpret_cast<const u32x4*>(smem + m * 512 + ((e_ln ^ (m & 7)) * 16));
    if ((m >> 3) & 1) x = u32x4{x[2], x[3], x[0], x[1]};
    const int n = n0 + e_ln * 8;
    if (m0 + m < a.M && n < a.N) pp_store_out(reinterpret_cast<u32x4*>(reinterpret_cast<half_t*>(a.C) + (long)(m0 + m) * a.N + n), x, a.ws_policy);
  }
#endif
}

// ------------------------------------------------------------------------------------------
// dense fp8, 128 x 128 tile: the member for outputs too small to give every CU a wider tile - the N / 8 column shards of
// BASELINE c5 (4096 x 1024: 128 tiles of 128 x 256, 256 of these).  Wave grid 4 (n) x 2 (m): a wave owns 32 weight rows and
// 64 activation rows (4 fragments x 2 = 8 MFMAs per k-tile: ONE phase), the two m-halves are the two role groups of the
// ping-pong (waves w and w + 4 share a SIMD) and each keeps its own copy of the 32 weight rows (the second fetch is an L2 hit).
// Everything else is wq_gemm_pp8_kernel's 128-row form: activation ring of 4, weight ring of 3 per wave, the k-tile three
// ahead issued as [two activation pieces, four weight pieces], one counted vmcnt(12) per k-tile.
// ------------------------------------------------------------------------------------------
template <int WFMT_, int AFMT_>
struct PP8SPolicy {
  static constexpr int WFMT = WFMT_, AFMT = AFMT_;
  static constexpr int BM = 128, BN = 128, THREADS = 512, KT = 128, TILE_ROW = 128;
  static constexpr int RING = 4, D = 3, WS = 3;
  static constexpr int A_SLOT = BM * TILE_ROW;
  static constexpr int W_OFF = RING * A_SLOT;
  static constexpr int LDS_BYTES = W_OFF + 8 * WS * 4096;
  static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= BM * BN * 2, "LDS budget / output staging");
};

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_pp8s_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int D = P::D, RING = P::RING, WS = P::WS;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave & 3, mg = wave >> 2;           // mg is also the role group
  const int fr = lane & 15, kb = lane >> 4;

  const TileOfBlock tob = tile_of_block(a, (int)blockIdx.x, (int)gridDim.x);
  const int tile_m = tob.tile_m, tile_n = tob.tile_n;
  const int m0 = tile_m * P::BM, n0 = (tile_n + a.tile_n_off) * P::BN;
  const int ntiles = a.K / P::KT;

  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((long)a.M * a.K), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.B), 0, (int)((long)a.N * a.K), 0x00020000);
  const int g0_ = (lane & 7) ^ ((lane >> 4) & 7);
  const uint32_t v0w = (uint32_t)(ng * 32 + (lane >> 3)) * (uint32_t)a.K + (uint32_t)(g0_ * 16);       // the wave's 32 weight rows
  const uint32_t v0a = (uint32_t)(wave * 16 + (lane >> 3)) * (uint32_t)a.K + (uint32_t)(g0_ * 16);     // its two pieces of the activation tile
  const int vd = ((g0_ ^ 4) - g0_) * 16;
  const uint32_t a_rows0 = (uint32_t)m0 * (uint32_t)a.K, w_rows0 = (uint32_t)n0 * (uint32_t)a.K;

  unsigned char* const a_ring = smem;
  unsigned char* const w_ring = smem + P::W_OFF + wave * (WS * 4096);
  auto dma_a = [&](int tt, int slot, int j) {
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = a_ring + slot * P::A_SLOT + (wave * 16 + j * 8) * P::TILE_ROW;
    const uint32_t rows = a_rows0 + (uint32_t)(j * 8) * (uint32_t)a.K;
    const uint32_t voff = (j & 1) ? v0a + (uint32_t)vd + rows : v0a + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };
  auto dma_w = [&](int tt, int j, int ws) {
    const int tc = tt < ntiles ? tt : ntiles - 1;
    unsigned char* dst = w_ring + ws * 4096 + j * 1024;
    const uint32_t rows = w_rows0 + (uint32_t)(j * 8) * (uint32_t)a.K;
    const uint32_t voff = (j & 1) ? v0w + (uint32_t)vd + rows : v0w + rows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)dst, 16, voff, tc * P::TILE_ROW, 0, 0);
  };

  const int swl = (fr >> 1) & 7;
  uint32_t rd[2];
  rd[0] = (uint32_t)(fr * P::TILE_ROW + ((kb ^ swl) * 16));
  rd[1] = (uint32_t)(fr * P::TILE_ROW + (((4 + kb) ^ swl) * 16));

  f32x4 acc[4][2];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) acc[f][nf] = f32x4{0, 0, 0, 0};
  u32x4 afrag[4][2], wfrag[2][2];

#pragma unroll
  for (int tt = 0; tt < 3; ++tt) {
    dma_a(tt, tt, 0);
    dma_a(tt, tt, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w(tt, j, tt);
  }
  pp_wait_vmcnt<12>();
  PP_BARRIER();
  if (mg == 1) PP_BARRIER();

  int slot = 0, wslot = 0;
  for (int t = 0; t < ntiles; ++t) {
    // ---- load segment ----
    {
      const unsigned char* sl = a_ring + slot * P::A_SLOT + mg * (64 * P::TILE_ROW);
      const unsigned char* ws = w_ring + wslot * 4096;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int i = 0; i < 2; ++i) wfrag[nf][i] = *reinterpret_cast<const u32x4*>(ws + nf * 2048 + rd[i]);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < 2; ++i) afrag[f][i] = *reinterpret_cast<const u32x4*>(sl + f * (16 * P::TILE_ROW) + rd[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the weight slot is refilled below: its fragments are in registers
      const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
      dma_a(t + D, dslot, 0);
      dma_a(t + D, dslot, 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_w(t + 3, j, wslot);
      pp_wait_vmcnt<12>();                           // tile t + 1 complete: only the two younger tiles (six operations each) outstanding
      PP_BARRIER();
    }
    // ---- compute segment ----
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const i32x8 av = {(int)afrag[f][0][0], (int)afrag[f][0][1], (int)afrag[f][0][2], (int)afrag[f][0][3],
                        (int)afrag[f][1][0], (int)afrag[f][1][1], (int)afrag[f][1][2], (int)afrag[f][1][3]};
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const i32x8 wv = {(int)wfrag[nf][0][0], (int)wfrag[nf][0][1], (int)wfrag[nf][0][2], (int)wfrag[nf][0][3],
                          (int)wfrag[nf][1][0], (int)wfrag[nf][1][1], (int)wfrag[nf][1][2], (int)wfrag[nf][1][3]};
        acc[f][nf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, acc[f][nf], P::WFMT, P::AFMT, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    }
    PP_BARRIER();
    slot = slot + 1 == RING ? 0 : slot + 1;
    wslot = wslot + 1 == WS ? 0 : wslot + 1;
  }
  if (mg == 0) PP_BARRIER();

  // ---- epilogue: the 128 x 128 tile through LDS (256-byte rows: 16 pairs of 8-byte units), four rows per store instruction ----
  pp_wait_vmcnt<0>();
  PP_BARRIER();
  const int el = pp_opaque(lane);
  const int e_fr = el & 15, e_kb = el >> 4;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int m = (mg * 4 + f) * 16 + e_fr;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const half2_t lo = {(half_t)acc[f][nf][0], (half_t)acc[f][nf][1]}, hi = {(half_t)acc[f][nf][2], (half_t)acc[f][nf][3]};
      const int u = ng * 8 + nf * 4 + e_kb;           // 8-byte unit of the row (4 consecutive n)
      const int up = (((u >> 1) ^ (m & 7)) << 1) | ((u & 1) ^ ((m >> 3) & 1));
      *reinterpret_cast<u32x2*>(smem + m * 256 + up * 8) = u32x2{as_u32(lo), as_u32(hi)};
    }
  }
  PP_FENCE();
  __syncthreads();
  const int e_c = el & 15, e_r = el >> 4;            // 16-byte pair of the row, row of the four
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int m = wave * 16 + rr * 4 + e_r;
    u32x4 x = *reinterpret_cast<const u32x4*>(smem + m * 256 + ((e_c ^ (m & 7)) * 16));
    if ((m >> 3) & 1) x = u32x4{x[2], x[3], x[0], x[1]};
    const int n = n0 + e_c * 8;
    if (m0 + m < a.M && n < a.N) pp_store_out(reinterpret_cast<u32x4*>(reinterpret_cast<half_t*>(a.C) + (long)(m0 + m) * a.N + n), x, a.ws_policy);
  }
#endif
}


}  // namespace wqaa
