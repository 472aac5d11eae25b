// mm_lab_kernel.h - LAB ONLY (tools/gemm_lab --kind mm): a mid-M W_q x A streaming member that was built, measured and NOT shipped -
// it ties with the shipped skinny members (DESIGN.md 3.2b has the time lines and the reason).  Kept as the harness the numbers in
// profiles/r03_lab_mm_*.txt come from.  The regime it targets: the reference's small-block heuristics
// (bitblas/ops/general_matmul/tilelang/dequantize/matmul_dequantize_mma.py:127-168) and split-K variant (general_matmul_splitk.py:26-89).
//
// At mid M a workgroup's share of the matrix pipe is a few microseconds, the weights are read once from HBM (a 1 - 2 us
// round trip) and the activations, though small, arrive first-touch from HBM too (every workgroup of an XCD walks K in step):
// the kernel is a latency chain, not a throughput loop.  What it is built around (tools/gemm_lab --kind mm, time lines
// in profiles/r03_lab_mm_*.txt):
//   * vmcnt retires IN ORDER, so a wave that issues both kinds of load gives the HBM stream only the look-ahead of the L2 one.
//     The loads are split by WAVE: waves 0-3 issue every activation k-tile (ring of RING slots, RING - 1 tiles in flight),
//     waves 4-7 every weight chunk (a chunk = the 128-byte line of each of the tile's 128 rows = four k-tiles; WB buffers,
//     WB - 1 chunks = 8 - 12 k-tiles in flight) and the Scale / Zeros windows; each kind is waited with its own counted
//     vmcnt and published by the one s_barrier per k-tile.  All bytes arrive by LDS-DMA, nothing is drained in the loop.
//   * tile BM x 128, 8 waves.  A wave owns NW 16-row weight fragments (decoded in registers, pp_decode_f16) and BM / NW
//     activation rows: at BM = 32 one fragment and every row (decode once per workgroup); from BM = 64 two fragments and
//     half the rows - each activation fragment read from LDS feeds two MFMAs, which is what keeps the LDS port (128 B/clk)
//     under the matrix pipe - at the price of decoding every weight in two waves.
//   * activation fragments are double buffered in REGISTERS: the reads for k-tile t + 1 are issued right after the barrier
//     and return while the MFMAs of k-tile t run.
//   * K is split across workgroups until the chip is full (fp32 partial sums + wq_splitk_reduce_kernel, as wq_gemm_kernel:
//     the order of the sum is fixed, the result does not depend on which workgroup finishes first); with ksplit == 1 the
//     kernel stores float16 itself.  Consecutive workgroups of an XCD share one k-slice: its activation band is fetched once.
#pragma once
#include "wqaa_gemm_pp_kernel.h"

namespace wqaa {

template <int KIND_, int LAYOUT_, int MODE_, int NW_, int MF_, int OPT_ = 0>
struct MMPolicy {
  static constexpr int KIND = KIND_, LAYOUT = LAYOUT_, AT = AT_F16, MODE = MODE_, FLAGS = 0, NW = NW_, MF = MF_, OPT = OPT_;
  static constexpr int GM = NW_, GN = 8 / NW_;           // wave grid: GN groups of 16 * NW weight rows x GM groups of 16 * MF activation rows
  static constexpr int BM = 16 * MF_ * GM, BN = 128, THREADS = 512, KT = 64, TILE_ROW = 128;
  static constexpr int A_SLOT = BM * TILE_ROW;
  static constexpr int RING = BM == 128 ? 6 : 8, D = RING - 1;
  static constexpr int APW = BM / 32;                    // activation pieces (8 rows x 128 B) each of the four loading waves issues per k-tile
  static constexpr int WB = BM == 128 ? 3 : 4;           // weight-chunk buffers (16 KiB each)
  static constexpr int W_BUF = 128 * 128;
  static constexpr int W_OFF = RING * A_SLOT;
  static constexpr int META_OFF = W_OFF + WB * W_BUF, META_BUF = 4096;
  static constexpr bool HAS_META = MODE_ != MD_NONE;
  static constexpr int LDS_BYTES = META_OFF + 2 * META_BUF;
  using T = KindTraits<KIND_, AT_F16>;
  static_assert(NW_ == 1 || NW_ == 2, "one or two weight fragments per wave");
  static_assert(BM == 32 || BM == 64 || BM == 128, "32-, 64- or 128-row tile");
  static_assert(T::BITS == 4, "4-bit weights");
  static_assert(WB <= 4, "the Scale / Zeros window is waited through the weight chunks issued after it");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class P>
__global__ void __launch_bounds__(P::THREADS) wq_gemm_mm_kernel(const GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = typename P::T;
  constexpr int MODE = P::MODE, NW = P::NW, MF = P::MF, D = P::D, RING = P::RING, APW = P::APW, WB = P::WB;
  constexpr bool ZP = MODE == MD_ZO || MODE == MD_ZR;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kb = lane >> 4;
  const int ng = NW == 2 ? (wave & 3) : wave, mg = NW == 2 ? (wave >> 2) : 0;
  const bool a_role = wave < 4;            // waves 0-3 load activations, waves 4-7 weights and Scale / Zeros
  const int w4 = wave & 3;
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int idx) {
    if constexpr (P::OPT & PPO_TRACE) tr[idx] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);

  // ---- workgroup -> (k-slice, N tile, M tile); an XCD gets a contiguous range: one k-slice (or few), neighbouring N tiles ----
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  const int ntile = a.tiles_m * a.tiles_n;
  const int split = blk / ntile;
  blk -= split * ntile;
  const int tile_m = blk % a.tiles_m, tile_n = blk / a.tiles_m;
  const int m0 = tile_m * P::BM, n0 = tile_n * P::BN, nw0 = n0 + ng * (16 * NW);
  const int nch_all = a.K / 256;                                     // weight chunks (four k-tiles) in K
  const int c_begin = (int)((long)split * nch_all / a.ksplit), c_end = (int)((long)(split + 1) * nch_all / a.ksplit);
  const int t_begin = c_begin * 4, t_end = c_end * 4;                 // this workgroup's k-tiles
  const int ntiles_all = a.K / P::KT;

  // ---- LDS-DMA sources (see wq_gemm_pp_kernel): rows past the end of a matrix read zeros (buffer bounds), nothing is clamped ----
  const int a_row_bytes = a.K * 2;
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)((long)a.M * a_row_bytes), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.B), 0, (int)((long)a.N * a.row_bytes), 0x00020000);
  unsigned char* const a_ring = smem;
  unsigned char* const w_bufs = smem + P::W_OFF;
  unsigned char* const meta = smem + P::META_OFF;
  auto dma_a = [&](int tt, int slot) {                                // (waves 0-3) this wave's pieces of k-tile tt
    const int tc = tt < ntiles_all ? tt : ntiles_all - 1;
    const int l = pp_opaque(lane);
    const int g0 = (l & 7) ^ (l >> 4);                               // source granule of an even piece: the reader's swizzle, applied here
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      const int pr = w4 * APW + i;                                     // rows [8 pr, 8 pr + 8) of the tile
      const int g = (APW == 1 ? (pr & 1) : (i & 1)) ? (g0 ^ 4) : g0;
      const uint32_t voff = (uint32_t)(m0 + pr * 8 + (l >> 3)) * (uint32_t)a_row_bytes + (uint32_t)(g * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(a_ring + slot * P::A_SLOT + pr * 1024), 16, voff, tc * P::TILE_ROW, 0, 0);
    }
  };
  auto dma_w = [&](int chunk, int buf) {                              // (waves 4-7) rows [32 w4, +32) of a chunk: four pieces of 8 rows
    const int cc = chunk < nch_all ? chunk : nch_all - 1;
    const int l = pp_opaque(lane);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 8 + (l >> 3);
      const int g = (l & 7) ^ (((r & 15) >> 1) & 7);
      const uint32_t voff = (uint32_t)(n0 + 32 * w4 + r) * (uint32_t)a.row_bytes + (uint32_t)(g * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(w_bufs + buf * P::W_BUF + (32 * w4 + p * 8) * 128), 16, voff, cc * 128, 0, 0);
    }
  };
  const uint32_t mlim = P::HAS_META ? (uint32_t)a.N * (uint32_t)a.kg - 8u : 0u;
  const int nbodies_all = ntiles_all >> 1;
  auto group_of_body = [&](int b) -> int {
    b = b < nbodies_all ? b : nbodies_all - 1;
    return b >> a.gq_shift;
  };
  auto dma_meta = [&](int q) {             // (waves 4-7) window q (8 groups) of rows [32 w4, +32): lanes 0-31 Scale, 32-63 Zeros -> buffer q & 1
    if constexpr (P::HAS_META) {
      const int l = pp_opaque(lane);
      if (ZP || l < 32) {
        const int n = n0 + 32 * w4 + (l & 31);
        const uint16_t* mbase = (ZP && l >= 32) ? reinterpret_cast<const uint16_t*>(a.zeros) : reinterpret_cast<const uint16_t*>(a.scale);
        const uint32_t e = (uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg + (uint32_t)(q * 8);
        const uint16_t* src = mbase + (e < mlim ? e : mlim);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (lds_ptr_t)(meta + (q & 1) * P::META_BUF + w4 * 1024), 16, 0, 0);
      }
    }
  };

  DecodeCtx cx;
  cx.zf = (a.is_signed && P::KIND != DK_LUT4) ? (half_t)8.0f : (half_t)0.0f;
  cx.flip = 0u;
  cx.off8 = (half_t)0.0f;
#pragma unroll
  for (int b = 0; b < 8; ++b) cx.magic[b] = (uint32_t)((25 - b) << 10) * 0x00010001u;
  asm volatile("" : "+v"(cx.magic[0]));
  asm volatile("" : "+v"(cx.magic[4]));
  Lut16 lut;
  if constexpr (P::KIND == DK_LUT4) {
    if (a.fp4_table) lut = make_fp4_lut(false);
    else lut = make_lut16(reinterpret_cast<const half_t*>(a.lut));
  }

  f32x4 acc[NW][MF];
#pragma unroll
  for (int nf = 0; nf < NW; ++nf)
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[nf][f] = f32x4{0, 0, 0, 0};
  u32x4 A[2][MF * 2];                      // activation fragments of a k-tile, double buffered: [buffer][fragment * 2 + MFMA of the tile]
  uint32_t bw[2][NW][2][4];                // decoded weight operands: [tile parity][fragment][MFMA of the tile]
  uint32_t rawc[4][NW][2];                 // packed words of the chunk in hand: [k-tile][fragment][MFMA of the tile]
  uint32_t rawn[NW][2];                    // ... and of the first k-tile of the next chunk (decoded during the last tile of this one)
  half2_t s2c[NW], zAc[NW], zBc[NW];
  uint32_t m_s[NW], m_z[NW];
  int mlim_f[NW];
#pragma unroll
  for (int nf = 0; nf < NW; ++nf) {
    rawn[nf][0] = rawn[nf][1] = 0u;
    s2c[nf] = splat((half_t)1.0f);
    zAc[nf] = zBc[nf] = splat((half_t)0.0f);
    m_s[nf] = m_z[nf] = 0u;
    mlim_f[nf] = 0;
    if constexpr (P::HAS_META) {
      const int n = nw0 + nf * 16 + fr;
      mlim_f[nf] = (int)(mlim - (uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg);
    }
  }
  auto meta_read = [&](int b) {
    if constexpr (P::HAS_META) {
      const int gi = group_of_body(b);
      const int q8 = gi & ~7;
      const int l = pp_opaque(lane);
#pragma unroll
      for (int nf = 0; nf < NW; ++nf) {
        const int row = ng * (16 * NW) + nf * 16 + (l & 15);           // row of the tile: window block row >> 5, slot row & 31
        const int e = gi - (q8 < mlim_f[nf] ? q8 : mlim_f[nf]);
        const unsigned char* p = meta + ((gi >> 3) & 1) * P::META_BUF + (row >> 5) * 1024 + (row & 31) * 16 + e * 2;
        m_s[nf] = *reinterpret_cast<const uint16_t*>(p);
        if constexpr (ZP) m_z[nf] = *reinterpret_cast<const uint16_t*>(p + 512);
      }
    }
  };
  auto meta_convert = [&](auto ZI) {
    if constexpr (P::HAS_META) {
#pragma unroll
      for (int nf = 0; nf < NW; ++nf) {
        s2c[nf] = splat(bits_to_half(m_s[nf]));
        if constexpr (ZP) {
          const half_t z = bits_to_half(m_z[nf]);
          if constexpr (decltype(ZI)::value) {
            zAc[nf] = splat((half_t)1024.0f + cx.zf + z);
            zBc[nf] = splat((half_t)64.0f + cx.zf + z);
          } else {
            zAc[nf] = splat(z);
          }
        }
      }
    }
  };
  auto decode = [&](auto ZI, int nf, uint32_t w, uint32_t (&out)[4]) {
    pp_decode_f16<P, decltype(ZI)::value != 0>(w, cx.zf, s2c[nf], zAc[nf], zBc[nf], cx, lut, out);
  };
  // the word of (row, k-tile tq, MFMA jj) is word kb of granule 2 tq + jj of the row's 128-byte line
  auto read_chunk = [&](int buf) {
    const int l = pp_opaque(lane);
    const int sw = ((l & 15) >> 1) & 7;
#pragma unroll
    for (int nf = 0; nf < NW; ++nf) {
      const unsigned char* wb = w_bufs + buf * P::W_BUF + (ng * (16 * NW) + nf * 16 + (l & 15)) * 128 + (l >> 4) * 4;
#pragma unroll
      for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) rawc[tq][nf][jj] = *reinterpret_cast<const uint32_t*>(wb + (((2 * tq + jj) ^ sw) * 16));
    }
  };
  auto read_first = [&](int buf) {
    const int l = pp_opaque(lane);
    const int sw = ((l & 15) >> 1) & 7;
#pragma unroll
    for (int nf = 0; nf < NW; ++nf) {
      const unsigned char* wb = w_bufs + buf * P::W_BUF + (ng * (16 * NW) + nf * 16 + (l & 15)) * 128 + (l >> 4) * 4;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) rawn[nf][jj] = *reinterpret_cast<const uint32_t*>(wb + ((jj ^ sw) * 16));
    }
  };
  auto read_frags = [&](int slot, u32x4 (&dst)[MF * 2]) {
    const int l = pp_opaque(lane);
    const int swl = ((l & 15) >> 1) & 7;
    const unsigned char* sl = a_ring + slot * P::A_SLOT + (mg * MF) * (16 * P::TILE_ROW) + (l & 15) * P::TILE_ROW;
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) dst[f * 2 + jj] = *reinterpret_cast<const u32x4*>(sl + f * (16 * P::TILE_ROW) + (((4 * jj + (l >> 4)) ^ swl) * 16));
  };

  stamp(1);

  // ---- prologue: the first window, WB weight chunks (waves 4-7) and D k-tiles (waves 0-3) in flight ----
  int q_loaded = P::HAS_META ? (group_of_body(t_begin >> 1) >> 3) : 0;
  if (a_role) {
#pragma unroll
    for (int tt = 0; tt < D; ++tt) dma_a(t_begin + tt, tt);
  } else {
    dma_meta(q_loaded);
#pragma unroll
    for (int c = 0; c < WB; ++c) dma_w(c_begin + c, c);
  }
  stamp(2);
  // ---- the integer-zero fast path is a wave's choice: every zero point of its rows (all of K) must be one
  //      (its loads queue behind the prologue's: by the time they are back the first tile is too) ----
  bool zint = false;
  if constexpr (MODE == MD_ZO && P::KIND == DK_INT4) {
    bool ok = true;
    const uint16_t* zrow = reinterpret_cast<const uint16_t*>(a.zeros);
    const int n = nw0 + (lane & (16 * NW - 1));
    const uint32_t rowbase = (uint32_t)(n < a.N ? n : a.N - 1) * (uint32_t)a.kg;
    for (int i = (lane / (16 * NW)) * 8; i < a.kg; i += 8 * (64 / (16 * NW))) {
      const uint32_t e = rowbase + (uint32_t)i;
      const u32x4 v = *reinterpret_cast<const u32x4*>(zrow + (e < mlim ? e : mlim));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float z = (float)bits_to_half(v[k >> 1] >> ((k & 1) * 16)) + (float)cx.zf;
        ok = ok && z == __builtin_truncf(z) && z > -48.f && z < 48.f;
      }
    }
    zint = __all(ok);
  }
  if (a_role) pp_wait_vmcnt<(D - 1) * APW>();   // k-tile 0 landed; the younger ones may stay in flight
  else pp_wait_vmcnt<4 * (WB - 1)>();           // the window and chunk 0 landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  PP_BARRIER();
  stamp(3);
  read_chunk(0);
  meta_read(t_begin >> 1);
  read_frags(0, A[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- main loop: one weight chunk (four k-tiles) per trip; register roles are compile-time, the loop exists once per
  //      (decode flavour, load role) so that nothing inside it branches ----
  int slot = 0;                            // ring slot of the k-tile in hand
  int wslot = 0;                           // buffer of the chunk in hand
  int t = t_begin;
  auto step = [&](auto ZI, auto AR, auto TQ) {
    constexpr int tq = decltype(TQ)::value;
    constexpr bool ar = decltype(AR)::value != 0;
    constexpr int cur = tq & 1, nxt = cur ^ 1, par = tq & 1;
    const int nwslot = wslot + 1 == WB ? 0 : wslot + 1;
    if constexpr (P::OPT & PPO_ABL_NODMA) {
    } else if constexpr (ar) {
      // the k-tile D ahead takes the slot k-tile t - 1 has left (its reads returned before the barrier every wave has passed since)
      const int dslot = slot + D >= RING ? slot + D - RING : slot + D;
      dma_a(t + D, dslot);
      pp_wait_vmcnt<(D - 1) * APW>();      // k-tile t + 1 complete (this wave's pieces): only the D - 1 tiles issued since may be outstanding
    } else {
      // the chunk WB ahead takes the buffer of the chunk in hand: its words went to registers a trip ago, and every wave has
      // passed a barrier since its reads returned.  Chunk c + 1 is first read in step 2: all but the WB - 1 youngest chunks done.
      if constexpr (tq == 1) dma_w((t >> 2) + WB, wslot);
      if constexpr (tq == 2) pp_wait_vmcnt<4 * (WB - 1)>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(P::OPT & PPO_ABL_NOBAR)) PP_BARRIER();
    // picked up a tile ahead of their decode, so their LDS latency hides behind this tile's MFMAs
    if constexpr (tq == 2) read_first(nwslot);
    if constexpr (tq == 3) read_chunk(nwslot);
    if constexpr ((tq & 1) == 0) meta_read((t + 2) >> 1);
    const int nslot = slot + 1 == RING ? 0 : slot + 1;
    if constexpr (!(P::OPT & PPO_ABL_NOREAD)) read_frags(nslot, A[nxt]);
    // operands of the next tile, decoded in the shadow of this tile's MFMAs
    if constexpr ((tq & 1) == 1) meta_convert(ZI);
#pragma unroll
    for (int nf = 0; nf < NW; ++nf)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if constexpr (P::OPT & PPO_ABL_NODEC) {
          bw[par ^ 1][nf][jj][0] = tq == 3 ? rawn[nf][jj] : rawc[(tq + 1) & 3][nf][jj];
        } else {
          decode(ZI, nf, tq == 3 ? rawn[nf][jj] : rawc[(tq + 1) & 3][nf][jj], bw[par ^ 1][nf][jj]);
        }
      }
    if constexpr (!(P::OPT & PPO_ABL_NOMFMA))
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int nf = 0; nf < NW; ++nf) {
          const u32x4 bv = {bw[par][nf][jj][0], bw[par][nf][jj][1], bw[par][nf][jj][2], bw[par][nf][jj][3]};
          acc[nf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, bv), __builtin_bit_cast(half8_t, A[cur][f * 2 + jj]), acc[nf][f], 0, 0, 0);
        }
    slot = nslot;
    ++t;
    if constexpr (tq == 3) wslot = nwslot;
  };
  auto main_loop = [&](auto ZI, auto AR) {
    for (; t < t_end;) {
      if constexpr (P::HAS_META && decltype(AR)::value == 0) {
        // the window after the one this trip reads: asked for when its predecessor comes into use (>= 4 trips ahead), so the
        // counted waits on the WB <= 4 chunks issued after it cover it
        const int qn = (group_of_body(t >> 1) >> 3) + 1;
        if (qn != q_loaded) {
          dma_meta(qn);
          q_loaded = qn;
        }
      }
      step(ZI, AR, ic<0>{});
      step(ZI, AR, ic<1>{});
      step(ZI, AR, ic<2>{});
      step(ZI, AR, ic<3>{});
    }
  };
  // first operands: tile t_begin's words
  if (zint) {
    meta_convert(ic<1>{});
#pragma unroll
    for (int nf = 0; nf < NW; ++nf)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) decode(ic<1>{}, nf, rawc[0][nf][jj], bw[0][nf][jj]);
    if (a_role) main_loop(ic<1>{}, ic<1>{});
    else main_loop(ic<1>{}, ic<0>{});
  } else {
    meta_convert(ic<0>{});
#pragma unroll
    for (int nf = 0; nf < NW; ++nf)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) decode(ic<0>{}, nf, rawc[0][nf][jj], bw[0][nf][jj]);
    if (a_role) main_loop(ic<0>{}, ic<1>{});
    else main_loop(ic<0>{}, ic<0>{});
  }

  // ---- output: accumulator (nf, f) holds activation row m0 + 16 (mg MF + f) + fr, weight rows nw0 + 16 nf + 4 kb + {0..3} ----
  stamp(4);
  pp_wait_vmcnt<0>();                      // (pieces past the end of K still write LDS; nothing reads them)
  stamp(5);
  auto trace_out = [&]() {
    if constexpr (P::OPT & PPO_TRACE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(6);
      if (lane == 0 && a.lut) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<void*>(a.lut)) + ((long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = tr[i];
      }
    }
  };
  if (a.ksplit > 1) {
    f32x4* ws = reinterpret_cast<f32x4*>(a.ws);
#pragma unroll
    for (int nf = 0; nf < NW; ++nf) {
      const int nb = nw0 + nf * 16 + kb * 4;
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = m0 + (mg * MF + f) * 16 + fr;
        if (m < a.M && nb < a.N) ws[(((long)split * a.M + m) * a.N + nb) >> 2] = acc[nf][f];
      }
    }
    trace_out();
    return;
  }
#pragma unroll
  for (int nf = 0; nf < NW; ++nf) {
    const int nb = nw0 + nf * 16 + kb * 4;
    half_t bias_h[4] = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if (a.has_bias && nb < a.N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bias_h[i] = reinterpret_cast<const half_t*>(a.bias)[nb + i];
    }
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      const int m = m0 + (mg * MF + f) * 16 + fr;
      if (m >= a.M || nb >= a.N) continue;
      half_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = (half_t)acc[nf][f][i];
        if (a.has_bias) v[i] = v[i] + bias_h[i];
      }
      const half2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
      *reinterpret_cast<u32x2*>(reinterpret_cast<half_t*>(a.C) + (long)m * a.N + nb) = u32x2{as_u32(lo), as_u32(hi)};
    }
  }
  trace_out();
#endif
}

}  // namespace wqaa
