// wqaa_gemvx_kernel.h - the exact-product GEMV members (sub-byte integer weights x fp16 activations, M <= 2).
//
// Same operator as wqaa_gemv_kernel.h (reference templates: bitblas/ops/general_matmul/tilelang/dequantize/
// gemv_dequantize_simt.py:83-262), for callers that do not ask for the TE definition's per-element rounding of the
// dequantised weight (`strict_reference = 0`).  The TE graph materialises B_decode[n, k] = (w - z) * s ROUNDED to
// float16 per element (tirscript/matmul_dequantize_impl.py:435-449) - two packed VALU operations per pair of weights
// next to the unpack and the dot, and on MI355X that instruction stream, not HBM, bounds the int4 GEMV (PMC,
// profiles/r02_pmc_before.json: 22.6 VALU per weight word, waves issue-stalled 55 % of their time at 4.2 TB/s).
// Here the products are exact:
//     C[m, n] = cast( sum_groups  s[n, g] * ( sum_{k in g} q[n, k] * A[m, k]  -  z[n, g] * sum_{k in g} A[m, k] ) )
//   * a b-bit field AND-ed out of the packed word IS a float16 denormal q * 2^-24 (bits 0..7 of a half are mantissa
//     bits): V_DOT2C_F32_F16 takes denormal inputs at face value (tools/denorm_probe.hip), so the unpack is one
//     V_AND_B32 per pair of weights - no magic exponent, no subtract, no multiply.  Fields at bit offset o of a byte
//     come out scaled by 2^o: one fp32 accumulator per offset class, combined once per 16-byte lane chunk;
//   * the zero point (2^(b-1) of the signed formats, Zeros, QZeros) multiplies the chunk's activation SUM, which is
//     computed once per workgroup while the activation tile is staged into LDS;
//   * the scale multiplies the fp32 partial sum of the lane chunk (a chunk never straddles a group).
// The result differs from the TE definition's by the float16 rounding it does NOT do (relative 2^-11 per element,
// ~2e-4 of the output rms at K = 4096), i.e. it is closer to the real-valued product; `strict_reference = 1` keeps the
// per-element rounding.
//
// Skeleton (shared with the strict family): one wave64 streams R weight rows with 16-byte non-temporal loads, D lane
// chunks in flight per row; activations staged once per workgroup into LDS in the order the unpack produces.  New:
//   * KW waves of a workgroup split K of the same rows (step i goes to wave i % KW) and meet in LDS in a fixed
//     order - for few rows x long K (N = 1024 ... 4096 per-rank shards of the multi-GPU split) this
//     is what keeps >= 64 KiB of loads in flight per CU (reference: bitblas/ops/general_matmul_splitk.py:27-199
//     splits K across kernels and sums with torch.sum);
//   * wave reduction by DPP row_bcast (6 operations, result in lane 63).
#pragma once
#include "wqaa_common.h"
#include "wqaa_decode.h"
#include "wqaa_kinds.h"

namespace wqaa {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct GemvxArgs {
  const void* A;
  const void* B;
  const void* scale;
  const void* zeros;
  const void* bias;
  void* C;
  int m, N, K;
  int kg;             // groups per weight row
  int gq_shift;       // lane chunks per group as a shift (-1: gq_magic)
  uint32_t gq_magic;
  int nc;             // lane chunks (64 lanes x 16 B) per weight row
  int cpr;            // valid 16-byte chunks per weight row
  int nsteps;         // ceil(nc / D)
  int kw;             // waves of a workgroup sharing a row group (K split), divides the workgroup's wave count
  long row_bytes;
  int has_bias, out_dtype;
  int zint;           // integer zero point folded into the format: 2^(bits-1) for signed, 1 for int1 (flipped)
  uint32_t flip;      // int1 signed: ~w
  int zq_row_bytes;
  int n_rgb;          // row-group blocks: ceil(ceil(N / R) / (waves / kw))
  int slots;          // waves / kw: row groups a workgroup works on at a time
  uint32_t kw_magic;  // ceil(2^16 / kw): x / kw == (x * kw_magic) >> 16 for x < 4096 (no integer division in the prologue)
  const void* residual;  // PRO members (WQAA_EPI_ADD_RESIDUAL): (m, N) float16 added to the float16 result; NULL: none
  const void* norm_weight;   // NORM members (WQAA_EPI_RMSNORM_INPUT): (K,) float16 weight of the RMSNorm in front of the operator
  float norm_eps, norm_inv_k;
};

// One launch serves up to kGemvxGroupMax INDEPENDENT operators of one tile configuration (wqaa_matmul_group: the q/k/v
// or gate/up projections of a decoder layer - same K and format, their own N and pointers): blockIdx.y names the
// operator, every operator gets gridDim.x workgroups (those beyond its row-group blocks leave at once).  The launch
// boundary (~1.3 us) and the load ramp / decode tail of a 4 us GEMV are paid once instead of per operator.  A single
// call is the group of one (gridDim.y = 1): the same kernels, the same code path.
constexpr int kGemvxGroupMax = 8;
struct GemvxGroupArgs {
  GemvxArgs p[kGemvxGroupMax];
};

// ABL_: ablation bits for tools/ (lab members only, never selected by the library): 1 = loads consumed by one XOR
// instead of the decode + dot, 2 = no activation staging / barrier, 4 = no wave reduction / store, 8 = no store,
// 64 = time line: s_memrealtime stamps per wave written through a.bias ([workgroup][wave][8]; tools/gemv_lab.hip; slot 7 = the
//      norm's sum known, NORM members)
// AREG_: the lane keeps the activations of its own lane chunks in registers (4-bit LOP3 weights, M = 1, K within one
// step): the LOP3 interleave puts consecutive elements 2j, 2j + 1 into the two halves of field j, so the natural-order
// activation dword j IS the partner of masked field j - no LDS tile, no staging pass, no barrier; the chunk's
// activation sum is taken from the same registers.
// PRO_: the caller's elementwise ops behind the GEMV folded in - members of their own, so that the plain members' code
// (and registers) stay what they are.
//   1  residual add (wqaa.h WQAA_EPI_ADD_RESIDUAL): the storing lane fetches the residual of its rows while the weights of
//      the row group stream and adds it to the rounded result
//   2  gate / up pair (wqaa_matmul_gate_up): the wave's two rows are row n of TWO operators (grp.p[0] = gate_proj,
//      grp.p[1] = up_proj: same shape and format, own pointers); it stores half(silu(gate_out)) * up_out - the gated
//      activation, evaluated once per output element by the lane that holds both sums
//   3  RMSNorm in front (WQAA_EPI_RMSNORM_INPUT): A is the layer's hidden state; the workgroup takes sum x^2 of every row over
//      the items it has just loaded, and stages weight * half(x * rsqrt(mean + eps)) - the reference's BitnetRMSNorm
//      (= LlamaRMSNorm) - as its activations.  One extra barrier and K / threads multiplies per thread in front of the stream.
//   4  3 + 2: the norm in front of a gate / up pair
template <int BITS_, int LAYOUT_, int MODE_, int MB_, int R_, int D_, int ABL_ = 0, bool AREG_ = false, int PRO_ = 0>
struct GemvxPolicy {
  static constexpr int BITS = BITS_, LAYOUT = LAYOUT_, MODE = MODE_, MB = MB_, R = R_, D = D_, ABL = ABL_;
  static constexpr bool AREG = AREG_;
  static constexpr bool PRO = PRO_ == 1, PAIR = PRO_ == 2 || PRO_ == 4, NORM = PRO_ >= 3;
  static_assert(!(AREG_ && PRO_ != 0), "the fused post ops come with the LDS-staged members");
  static_assert(!PAIR || R_ == 2, "a gate / up pair is the two rows of a wave");
  static_assert(!AREG_ || (BITS_ == 4 && LAYOUT_ == LAYOUT_LOP3 && MB_ == 1), "register-resident activations: 4-bit LOP3 weights, M = 1");
  // activation items per thread in flight ahead of the weight stream: 8 waves x 3 cover K = 12288 at 4 bit (rounds past
  // the tile are skipped wave-uniformly; 4096x11008 8.5 -> 7.76 us against one item).  The two-row members serve the
  // many-row shapes, where K is short and the extra registers cost 4 % (11008x4096 6.7 -> 7.0 us): they keep one.
  // (the norm needs the whole row in registers before it can stage anything: two items for the two-row members, K <= 8192 at 4 bit)
  static constexpr int NAI = R_ == 1 ? 3 : NORM ? 2 : 1;
  static constexpr int KIND = BITS_ == 4 ? DK_INT4 : BITS_ == 2 ? DK_INT2 : DK_INT1;
  using T = KindTraits<KIND, AT_F16>;
  static constexpr int EPW = 32 / BITS_;       // fields per 32-bit word
  static constexpr int NPAIR = EPW / 2;        // packed-half registers per word
  static constexpr int NCLS = 8 / BITS_;       // bit-offset classes inside a byte: scale 2^(BITS * c)
  static constexpr int E = 128 / BITS_;        // elements per 16-byte lane chunk
  static constexpr int PIECES = E / 8;         // 16-byte activation pieces per lane chunk
  static constexpr int PPW = EPW / 8;          // activation pieces per weight word
};

// torch's `F.silu(gate) * up` on float16 values: silu in fp32 (x / (1 + exp(-x))), rounded to float16, times up (a float16
// product is the correctly rounded exact product, which is what the fp32 multiply + cast of torch gives)
__device__ __forceinline__ half_t silu_mul_h(half_t g, half_t u) {
  const float gf = (float)g;
  return (half_t)(gf / (1.f + expf(-gf))) * u;
}

// wave64 sum, result valid in lane 63 (classic GCN row_bcast ladder: 6 DPP adds)
__device__ __forceinline__ float wave_sum_l63(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xE, true));   // row_shr:4 bank_mask:0xe
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xC, true));   // row_shr:8 bank_mask:0xc
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));   // row_bcast:15 row_mask:0xa
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));   // row_bcast:31 row_mask:0xc
  return v;
}

template <class P>
__global__ void __launch_bounds__(1024) wq_gemvx_kernel(const GemvxGroupArgs grp) {
  using T = typename P::T;
  const GemvxArgs a = grp.p[blockIdx.y];         // kernel-argument segment, indexed by a dispatch-time scalar
  constexpr int R = P::R, MB = P::MB, D = P::D, MODE = P::MODE, BITS = P::BITS;
  constexpr int EPW = P::EPW, NPAIR = P::NPAIR, NCLS = P::NCLS, E = P::E, PIECES = P::PIECES, PPW = P::PPW;
  constexpr int ZPB = 8 / BITS;                  // quantized zero points per byte

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int NW = nthreads >> 6;
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i) {
    if constexpr ((P::ABL & 64) != 0) tr_[i] = __builtin_amdgcn_s_memrealtime();
  };
  auto trace_out = [&]() {
    if constexpr ((P::ABL & 64) != 0) {
      if (lane == 0 && a.bias) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<void*>(a.bias)) +
                                  (((long)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = tr_[i];
      }
    }
  };
  stamp(0);
  // The prologue is a latency chain in front of the first load (kernel-argument fetch -> address arithmetic -> issue)
  // that nothing overlaps with: fetch every argument it needs in ONE scalar round trip (the compiler otherwise fetches
  // them where first used, three dependent s_load waits before the weight loads), and keep integer divisions out of
  // it (the host passes slots and a reciprocal of kw).
  asm volatile("" ::"s"(a.A), "s"(a.B), "s"(a.scale), "s"(a.zeros), "s"(a.N), "s"(a.K), "s"(a.kg), "s"(a.gq_shift), "s"(a.cpr),
               "s"(a.nsteps), "s"(a.kw), "s"(a.row_bytes), "s"(a.n_rgb), "s"(a.slots), "s"(a.kw_magic), "s"(a.m));
  if constexpr (P::PAIR) asm volatile("" ::"s"(grp.p[1].B), "s"(grp.p[1].scale), "s"(grp.p[1].zeros));
  if constexpr (P::NORM) asm volatile("" ::"s"(a.norm_weight), "s"(a.norm_eps), "s"(a.norm_inv_k));
  if constexpr (P::PRO) asm volatile("" ::"s"(a.residual));
  const int kw = a.kw;
  const int slots = a.slots;                     // row groups the workgroup works on at a time
  const int rgl = (int)(((uint32_t)wave * a.kw_magic) >> 16), kpart = wave - rgl * kw;
  const int nsteps = a.nsteps;
  const int ncp = nsteps * D;                    // chunk slots (zero activations beyond nc)
  const int nmy = (int)(((uint32_t)(nsteps - kpart + kw - 1) * a.kw_magic) >> 16);   // steps of a row group that fall to this wave (>= 1: kw <= nsteps)
  // LDS: activation pieces [mi][chunk][piece][lane] (16 B), chunk sums [mi][chunk][lane] (4 partials, 16 B), K-split partials
  u32x4* a_lds = reinterpret_cast<u32x4*>(smem_raw);
  float* sa_lds = reinterpret_cast<float*>(a_lds + (long)MB * ncp * PIECES * 64);
  float* red_lds = sa_lds + (long)MB * ncp * 64 * 4;                               // [2][slot][kpart][R * MB]

  // operand pointers of row r of a row group: one operator's - or, PAIR, row r = 0 of grp.p[0] (gate) and r = 1 of grp.p[1] (up)
  const uint8_t* Bp[R];
  const uint16_t* Sp[R];
  const uint16_t* Zp[R];
  const uint8_t* Qp[R];
  const void* biasp[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const GemvxArgs& o = (P::PAIR && r == 1) ? grp.p[1] : a;
    Bp[r] = reinterpret_cast<const uint8_t*>(o.B);
    Sp[r] = reinterpret_cast<const uint16_t*>(o.scale);
    Zp[r] = reinterpret_cast<const uint16_t*>(o.zeros);
    Qp[r] = reinterpret_cast<const uint8_t*>(o.zeros);
    biasp[r] = o.bias;
  }
  constexpr int RS = P::PAIR ? 1 : R;             // output rows a row group stands for
  const int n_rg = (a.N + RS - 1) / RS;
  auto row_of = [&](int rg, int r) { return P::PAIR ? rg : rg * R + r; };

  // this workgroup's row-group blocks rb.first, rb.first + rb.stride, ... < rb.end (XCD-aware, wqaa_kinds.h; many small
  // workgroups and the hardware dispatcher balance better than one persistent workgroup per CU: measured)
  const RowBlocks rb = xcd_row_blocks((int)blockIdx.x, (int)gridDim.x, a.n_rgb);
  if (rb.first >= rb.end) return;                              // grid padding / a shorter operator of a group: nothing to load
  int iters = 1;                                               // uniform over the workgroup; the usual case: one block, no division
  if (rb.first + rb.stride < rb.end) iters = (rb.end - rb.first + rb.stride - 1) / rb.stride;
  const int total = iters * nmy;                               // (row group, step) positions of this wave

  struct Stage {
    u32x4 w[R];
    uint32_t s[R], z[R];
  };
  auto rg_of = [&](int it) { return (rb.first + it * rb.stride) * slots + rgl; };
  // weight loads of chunk d of position (it, si) = (it-th row-group block of this workgroup, si-th own step): unconditional
  auto issue = [&](Stage& st, int it, int si, int d) {
    int rg = rg_of(it);
    rg = rg < n_rg ? rg : n_rg - 1;                 // a clamped slot re-reads the last row group and never stores
    int chunk = ((kpart + si * kw) * D + d) * 64 + lane;
    chunk = chunk < a.cpr ? chunk : 0;              // clamped lanes meet zero activations
    int gi = 0;
    if (MODE != MD_NONE) gi = a.gq_shift >= 0 ? (chunk >> a.gq_shift) : (int)__umulhi((uint32_t)chunk, a.gq_magic);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int n = row_of(rg, r);
      n = n < a.N ? n : a.N - 1;
      st.w[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Bp[r] + (long)n * a.row_bytes + (long)chunk * 16));
      if constexpr (MODE != MD_NONE) st.s[r] = Sp[r][(long)n * a.kg + gi];
      if constexpr (MODE == MD_ZO || MODE == MD_ZR) st.z[r] = Zp[r][(long)n * a.kg + gi];
      if constexpr (MODE == MD_ZQ) st.z[r] = Qp[r][(long)gi * a.zq_row_bytes + n / ZPB];
    }
  };

  // ---- activations first: their loads (L2 hits after the first workgroups) must not queue behind the weight stream -
  // loads return in order, so an activation load issued after the weights would only be usable after them.
  // Item = the EPW activations of one weight word of one lane chunk (4 VGPRs at 4 bit): small, so that the kernel
  // stays at <= 64 VGPRs - occupancy is what hides the LDS and VALU latencies of the decode (an item of a whole lane
  // chunk cost 87 VGPRs and 10-30 % of the throughput) ----
  constexpr int IVW = EPW / 8;                    // 16-byte vectors per item
  const int items = MB * ncp * 4 * 64;
  constexpr int NAI = P::NAI;                     // items per thread loaded ahead of the weights (8 waves x 1 item cover K = 4096 at 4 bit)
  u32x4 araw[NAI][IVW];
  auto item_src = [&](int idx, bool& valid) -> const u32x4* {
    const int l = idx & 63;
    const int u = (idx >> 6) & 3;
    int c = idx >> 8, mi = 0;                      // idx < items = MB * ncp * 256 wherever the result is used; MB <= 2
    if (MB > 1 && c >= ncp) { c -= ncp; mi = 1; }
    const int chunk = c * 64 + l;
    valid = idx < items && chunk < a.cpr && mi < a.m;
    return reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(a.A) +
                                          ((long)(valid ? mi : 0) * a.K + (long)(valid ? chunk : 0) * E + u * EPW) * 2);
  };
  auto item_store = [&](int idx, const u32x4 (&raw)[IVW], bool valid) {
    const int l = idx & 63;
    const int u = (idx >> 6) & 3;
    int c = idx >> 8, mi = 0;                      // idx < items = MB * ncp * 256 wherever the result is used; MB <= 2
    if (MB > 1 && c >= ncp) { c -= ncp; mi = 1; }
    float sum = 0.f;
    half_t el[EPW];
#pragma unroll
    for (int e = 0; e < EPW / 2; ++e) {
      const half2_t h = as_h2(raw[e / 4][e % 4]);
      el[2 * e] = h[0];
      el[2 * e + 1] = h[1];
      sum = __builtin_amdgcn_fdot2(h, half2_t{(half_t)1.f, (half_t)1.f}, sum, false);
    }
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2_t h = {el[T::src_elem(P::LAYOUT, pp * 8 + 2 * e)], el[T::src_elem(P::LAYOUT, pp * 8 + 2 * e + 1)]};
        out[e] = valid ? as_u32(h) : 0u;
      }
      a_lds[((long)(mi * ncp + c) * PIECES + u * PPW + pp) * 64 + l] = out;
    }
    sa_lds[((mi * ncp + c) * 64 + l) * 4 + u] = valid ? sum : 0.f;          // the chunk's sum arrives as four partials
  };
  bool avalid[NAI];
  u32x4 nraw[P::NORM ? NAI : 1][IVW];             // NORM: the norm weight's halves of the same items
  // register-resident activations (AREG): word u of lane chunk d pairs with areg[d][u]; sa_reg[d] = the chunk's sum
  u32x4 areg[D][4];
  float sa_reg[D];
  if constexpr (P::AREG) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int chunk = d * 64 + lane;                 // one step, no K split: chunk d of the row
      const bool valid = chunk < a.cpr;
      const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(a.A) + (long)(valid ? chunk : 0) * (E * 2));
#pragma unroll
      for (int u = 0; u < 4; ++u) areg[d][u] = src[u];
      if (!valid) {
#pragma unroll
        for (int u = 0; u < 4; ++u) areg[d][u] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  if constexpr (!(P::ABL & 2) && !P::AREG) {
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
      const u32x4* src = item_src(j * nthreads + tid, avalid[j]);
      if (j * nthreads < items) {                  // wave-uniform: whole rounds beyond the tile are skipped
#pragma unroll
        for (int v = 0; v < IVW; ++v) araw[j][v] = src[v];
        if constexpr (P::NORM) {
          // same elements of the (K,) weight: the item's address without its row offset
          const int idx = j * nthreads + tid;
          int c = idx >> 8;
          if (MB > 1 && c >= ncp) c -= ncp;
          const int chunk = c * 64 + (idx & 63);
          const u32x4* wsrc = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(a.norm_weight) +
                                                             ((long)(avalid[j] ? chunk : 0) * E + ((idx >> 6) & 3) * EPW) * 2);
#pragma unroll
          for (int v = 0; v < IVW; ++v) nraw[j][v] = wsrc[v];
        }
      }
    }
  }

  // ---- the first weight step, behind the activations ----
  Stage st[D];
#pragma unroll
  for (int d = 0; d < D; ++d) issue(st[d], 0, 0, d);
  stamp(1);

  if constexpr (P::AREG) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // the same additions in the same order as the LDS-staged members (item_store: one partial per weight word, then
      // (p0 + p1) + (p2 + p3)): which member a row runs through must not show in its bits (groups fuse members of both kinds)
      float part[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        part[u] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[u] = __builtin_amdgcn_fdot2(as_h2(areg[d][u][e]), half2_t{(half_t)1.f, (half_t)1.f}, part[u], false);
      }
      sa_reg[d] = (part[0] + part[1]) + (part[2] + part[3]);
    }
  }
  if constexpr (P::NORM) {
    // sum x^2 of every activation row: per thread over its items (fp32; the squares of float16 values are exact), per wave by
    // the DPP ladder, across the waves through LDS in wave order - every thread ends up with the same bits
    float ssq[MB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) ssq[mi] = 0.f;
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
      const int idx = j * nthreads + tid;
      float part = 0.f;
      if (idx < items && avalid[j]) {
#pragma unroll
        for (int v = 0; v < IVW; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e) part = __builtin_amdgcn_fdot2(as_h2(araw[j][v][e]), as_h2(araw[j][v][e]), part, false);
      }
      const bool second = MB > 1 && (idx >> 8) >= ncp;          // the item belongs to activation row 1
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) ssq[mi] += (second == (mi == 1)) ? part : 0.f;
    }
    float* nred = red_lds;                                      // free until the first `finish` (behind the staging barrier)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      const float w = wave_sum_l63(ssq[mi]);
      if (lane == 63) nred[mi * NW + wave] = w;
    }
    __syncthreads();
    float rinv[MB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      float tot = nred[mi * NW];
      for (int w = 1; w < NW; ++w) tot += nred[mi * NW + w];
      rinv[mi] = rsqrtf(tot * a.norm_inv_k + a.norm_eps);
    }
    stamp(7);                                                   // (lab time line: the norm's sum is known)
    // x -> weight * half(x * r): the two roundings of `self.weight * (x.float() * rsqrt(var + eps)).to(half)`
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
      const int idx = j * nthreads + tid;
      const float r = (MB > 1 && (idx >> 8) >= ncp) ? rinv[MB - 1] : rinv[0];
#pragma unroll
      for (int v = 0; v < IVW; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t x = as_h2(araw[j][v][e]);
          const half2_t h = {(half_t)((float)x[0] * r), (half_t)((float)x[1] * r)};
          araw[j][v][e] = as_u32(as_h2(nraw[j][v][e]) * h);
        }
    }
  }
  if constexpr (!(P::ABL & 2) && !P::AREG) {
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
      const int idx = j * nthreads + tid;
      if (idx < items) item_store(idx, araw[j], avalid[j]);
    }
    for (int idx = NAI * nthreads + tid; idx < items; idx += nthreads) {      // very long K: the rest queues behind the weights
      bool valid;
      const u32x4* src = item_src(idx, valid);
      u32x4 raw[IVW];
#pragma unroll
      for (int v = 0; v < IVW; ++v) raw[v] = src[v];
      item_store(idx, raw, valid);
    }
    __syncthreads();
  }
  stamp(2);
  if (total <= 0) return;

  const float zint = (float)a.zint;
  float acc[R][MB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) acc[r][mi] = 0.f;

  // one lane chunk of R rows against MB activation rows
  auto consume = [&](const Stage& s, int c, int rg_now, auto dc) {
    constexpr int DC = decltype(dc)::value;         // chunk index inside the step (register-resident activations)
    if constexpr (P::ABL & 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][0] += __builtin_bit_cast(float, (s.w[r][0] ^ s.w[r][1] ^ s.w[r][2] ^ s.w[r][3] ^ s.s[r]) & 0x3fffffu);
      return;
    }
    float cls[R][MB][NCLS];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) cls[r][mi][k] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                   // the chunk's four weight words
      uint32_t f[R][NPAIR];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t w = BITS == 1 ? (s.w[r][u] ^ a.flip) : s.w[r][u];      // (only the signed 1-bit format flips its codes)
        const uint32_t w8 = w >> 8;
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
          // pair i = the field at bit BITS*i of each 16-bit half: a float16 denormal q * 2^(o - 24), o = (BITS*i) & 7
          constexpr uint32_t fmask = ((1u << BITS) - 1u) * 0x00010001u;
          const int bit = BITS * i;
          f[r][i] = (bit >= 8 ? w8 : w) & (fmask << (bit & 7));
        }
      }
#pragma unroll
      for (int pp = 0; pp < PPW; ++pp) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          u32x4 av;
          if constexpr (P::AREG) av = areg[DC][u];
          else av = a_lds[((long)(mi * ncp + c) * PIECES + u * PPW + pp) * 64 + lane];
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = pp * 4 + e;               // pair index inside the word
              const int k = ((BITS * i) & 7) / BITS;   // offset class
              cls[r][mi][k] = __builtin_amdgcn_fdot2(as_h2(f[r][i]), as_h2(av[e]), cls[r][mi][k], false);
            }
        }
      }
    }
    // chunk epilogue: combine the classes, remove the zero point, apply the group scale
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      float sa;
      if constexpr (P::AREG) {
        sa = sa_reg[DC];
      } else {
        const f32x4_t sa4 = reinterpret_cast<const f32x4_t*>(sa_lds)[(mi * ncp + c) * 64 + lane];
        sa = (sa4[0] + sa4[1]) + (sa4[2] + sa4[3]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        // class k holds sum q * a * 2^(BITS * k - 24): Horner towards class 0, then undo the 2^-24 (all exact scalings)
        float t = cls[r][mi][NCLS - 1];
#pragma unroll
        for (int k = NCLS - 2; k >= 0; --k) t = __builtin_fmaf(t, 1.f / (float)(1 << BITS), cls[r][mi][k]);
        t *= 16777216.f;                                                   // sum q * a
        float z = zint;
        if constexpr (MODE == MD_ZO) z += (float)bits_to_half(s.z[r]);
        if constexpr (MODE == MD_ZQ) {
          const int n = row_of(rg_now, r);
          z = (float)((s.z[r] >> ((n % ZPB) * BITS)) & ((1u << BITS) - 1u));    // integer-domain zero: ignores signedness
        }
        t = __builtin_fmaf(-z, sa, t);
        if constexpr (MODE == MD_NONE) {
          acc[r][mi] += t;
        } else {
          acc[r][mi] = __builtin_fmaf(t, (float)bits_to_half(s.s[r]), acc[r][mi]);
          if constexpr (MODE == MD_ZR) acc[r][mi] = __builtin_fmaf(-(float)bits_to_half(s.z[r]), sa, acc[r][mi]);
        }
      }
    }
  };

  // PRO: the residual of this wave's rows, asked for when a row group starts (every lane the same address: one request),
  // so that the add in `finish` does not wait on memory
  float resv[R][MB];
  auto load_residual = [&](int it) {
    if constexpr (P::PRO) {
      const int rg = rg_of(it);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          const int n = rg * R + r;
          const bool ok = a.residual != nullptr && kpart == 0 && n < a.N && mi < a.m;   // the storing wave only (residual may alias C)
          resv[r][mi] = ok ? (float)reinterpret_cast<const half_t*>(a.residual)[(long)mi * a.N + n] : 0.f;
        }
    }
  };

  // a row group is complete for this wave: reduce over the wave, meet the kw - 1 other waves sharing the rows, store
  auto finish = [&](int it) {
    float tot[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) {
        tot[r][mi] = (P::ABL & 4) ? acc[r][mi] : wave_sum_l63(acc[r][mi]);
        acc[r][mi] = 0.f;
      }
    const int rg_done = rg_of(it);
    if (kw > 1) {
      float* red = red_lds + (it & 1) * (NW * R * MB);       // double buffered: one barrier per row group
      if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) red[(rgl * kw + kpart) * (R * MB) + r * MB + mi] = tot[r][mi];
      }
      __syncthreads();
      if (kpart == 0 && lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            float v = red[(rgl * kw) * (R * MB) + r * MB + mi];
            for (int p = 1; p < kw; ++p) v += red[(rgl * kw + p) * (R * MB) + r * MB + mi];   // fixed order
            tot[r][mi] = v;
          }
      }
    }
    if constexpr ((P::ABL & 48) != 0 && R == 2 && MB == 1) {
      // lab: store variants (float16 output, no bias): 16 = write-through (sc1) stores, 32 = the two rows as one dword
      if (kpart == 0 && lane == 63 && rg_done < n_rg) {
        half_t* cp = reinterpret_cast<half_t*>(a.C) + rg_done * 2;
        const half_t h0 = (half_t)tot[0][0], h1 = (half_t)tot[1][0];
        if constexpr ((P::ABL & 32) != 0) {
          const half2_t hh = {h0, h1};
          if constexpr ((P::ABL & 16) != 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(cp), as_u32(hh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *reinterpret_cast<uint32_t*>(cp) = as_u32(hh);
        } else {
          __hip_atomic_store(reinterpret_cast<uint16_t*>(cp), __builtin_bit_cast(uint16_t, h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(reinterpret_cast<uint16_t*>(cp) + 1, __builtin_bit_cast(uint16_t, h1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      return;
    }
    if constexpr (P::PAIR) {
      // float16 (the host checks): both projections' results (+ their biases) rounded to float16 as their own launches would
      // store them, then torch's `F.silu(gate) * up` on the two values
      if (kpart == 0 && lane == 63 && rg_done < n_rg) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          if (mi >= a.m) continue;
          half_t h[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            h[r] = (half_t)tot[r][mi];
            if (a.has_bias) h[r] = h[r] + reinterpret_cast<const half_t*>(biasp[r])[rg_done];
          }
          reinterpret_cast<half_t*>(a.C)[(long)mi * a.N + rg_done] = silu_mul_h(h[0], h[1]);
        }
      }
      return;
    }
    if constexpr (P::PRO) {
      // float16 output (the host checks): result (+ bias) rounded to float16, then + residual in fp32, rounded again
      if (kpart == 0 && lane == 63 && rg_done < n_rg) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          if (mi >= a.m) continue;
          half_t h[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int n = rg_done * R + r;
            h[r] = (half_t)tot[r][mi];
            if (a.has_bias && n < a.N) h[r] = h[r] + reinterpret_cast<const half_t*>(a.bias)[n];
            if (a.residual) h[r] = (half_t)((float)h[r] + resv[r][mi]);
          }
          const long idx = (long)mi * a.N + (long)rg_done * R;
          if (R == 2 && rg_done * R + 1 < a.N && !(idx & 1)) {
            *reinterpret_cast<uint32_t*>(reinterpret_cast<half_t*>(a.C) + idx) = as_u32(half2_t{h[0], h[R - 1]});
          } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
              if (rg_done * R + r < a.N) reinterpret_cast<half_t*>(a.C)[idx + r] = h[r];
          }
        }
      }
      return;
    }
    if (kpart == 0 && lane == 63 && rg_done < n_rg && (!(P::ABL & 12) || tot[0][0] == 123.f)) {
      if constexpr (R == 2) {
        // the two rows of the group are neighbours in C: ONE 4- / 8-byte store instead of two 2- / 4-byte ones
        // (same-call A/B: 4096^2 4.44 -> 4.27 us, 11008x4096 7.0 -> 6.8; a write-through (sc1) store costs 0.7 us)
        const int n = rg_done * 2;
        if (n + 1 < a.N && (a.out_dtype == WQAA_F16 || a.out_dtype == WQAA_F32)) {
          bool done = true;
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            if (mi >= a.m) continue;
            const long idx = (long)mi * a.N + n;
            if (idx & 1) { done = false; continue; }
            float b0 = 0.f, b1 = 0.f;
            if (a.has_bias) {
              b0 = (float)reinterpret_cast<const half_t*>(a.bias)[n];
              b1 = (float)reinterpret_cast<const half_t*>(a.bias)[n + 1];
            }
            if (a.out_dtype == WQAA_F16) {
              half_t h0 = (half_t)tot[0][mi], h1 = (half_t)tot[1][mi];
              if (a.has_bias) { h0 = h0 + (half_t)b0; h1 = h1 + (half_t)b1; }
              *reinterpret_cast<uint32_t*>(reinterpret_cast<half_t*>(a.C) + idx) = as_u32(half2_t{h0, h1});
            } else {
              float2_t v = {tot[0][mi], tot[1][mi]};
              if (a.has_bias) { v[0] += b0; v[1] += b1; }
              *reinterpret_cast<float2_t*>(reinterpret_cast<float*>(a.C) + idx) = v;
            }
          }
          if (done) return;
          // odd N with a second batch row: that row's pair straddles a 4-byte boundary - element stores below
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            if (mi >= a.m || !(((long)mi * a.N + n) & 1)) continue;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              float b = 0.f;
              if (a.has_bias) b = (float)reinterpret_cast<const half_t*>(a.bias)[n + r];
              store_out(a.C, (long)mi * a.N + n + r, tot[r][mi], a.out_dtype, a.has_bias != 0, b);
            }
          }
          return;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = rg_done * R + r;
        if (n >= a.N) continue;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
          if (mi >= a.m) continue;
          float b = 0.f;
          if (a.has_bias) b = (float)reinterpret_cast<const half_t*>(a.bias)[n];
          store_out(a.C, (long)mi * a.N + n, tot[r][mi], a.out_dtype, a.has_bias != 0, b);
        }
      }
    }
  };

  // positions in order; a position's weight registers are refilled with the next position's loads once all its chunks
  // have been consumed (65 VGPRs: 7 waves per SIMD - a deeper per-wave pipeline (activations prefetched into a
  // register ring, two steps of weights in flight: 105 VGPRs) measured 20-35 % SLOWER on every shape, occupancy wins)
  int it = 0, si = 0;
  for (int q = 0; q < total; ++q) {
    const int rg_now = rg_of(it);
    if constexpr (P::PRO) {
      if (si == 0) load_residual(it);
    }
    int c = (kpart + si * kw) * D;
    int it2 = it, si2 = si + 1;
    if (si2 == nmy) { si2 = 0; ++it2; }
    static_assert(D == 2, "two lane chunks per step");
    {
      if constexpr ((P::ABL & 64) != 0) {
        if (q == 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp(3);
        }
      }
      consume(st[0], c, rg_now, std::integral_constant<int, 0>{});
      ++c;
      // order fence by DATA dependence: the next chunk's LDS addresses and the next position's global addresses are
      // made to depend on this chunk's result.  Left alone hipcc hoists every chunk's LDS reads AND the next position's
      // global loads (renamed into a second register set) above the decode - 93 VGPRs, 5 waves per SIMD instead of 7;
      // sched_barrier / an asm memory clobber do not stop it (the loads are from memory it has proven read-only)
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[R - 1][0]), "+v"(acc[0][MB - 1]), "+v"(acc[R - 1][MB - 1]), "+s"(c), "+s"(it2), "+s"(si2));
    }
    {
      consume(st[1], c, rg_now, std::integral_constant<int, 1>{});
      ++c;
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[R - 1][0]), "+v"(acc[0][MB - 1]), "+v"(acc[R - 1][MB - 1]), "+s"(c), "+s"(it2), "+s"(si2));
    }
    if (q + 1 < total) {
#pragma unroll
      for (int d = 0; d < D; ++d) issue(st[d], it2, si2, d);
    }
    if constexpr ((P::ABL & 64) != 0) {
      if (q == 0) stamp(4);
      if (q + 1 == total) stamp(5);
    }
    if (si2 == 0) finish(it);
    it = it2;
    si = si2;
  }
  if constexpr ((P::ABL & 64) != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(6);
    trace_out();
  }
}

typedef void (*gemvx_fn)(const GemvxGroupArgs);

// member tables: wqaa_gemvx_inst_*.hip.  rd code = R * 10 + D
template <int BITS, int LAYOUT, int MODE, int MB>
static gemvx_fn pick_gemvx_pro_rd(int rd) {
  switch (rd) {
    case 12: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 1, 2, 0, false, 1>>;
    case 22: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2, 0, false, 1>>;
    case 1022: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2, 0, false, 2>>;     // gate / up pair
  }
  return nullptr;
}
template <int BITS, int LAYOUT>
static gemvx_fn pick_gemvx_pro_mode(int mode, int mb, int rd) {
#define WQAA_GXP(MODE) (mb == 1 ? pick_gemvx_pro_rd<BITS, LAYOUT, MODE, 1>(rd) : mb == 2 ? pick_gemvx_pro_rd<BITS, LAYOUT, MODE, 2>(rd) : nullptr)
  switch (mode) {
    case MD_NONE: return WQAA_GXP(MD_NONE);
    case MD_S: return WQAA_GXP(MD_S);
    case MD_ZO: return WQAA_GXP(MD_ZO);
    case MD_ZR: return WQAA_GXP(MD_ZR);
    case MD_ZQ: return WQAA_GXP(MD_ZQ);
  }
#undef WQAA_GXP
  return nullptr;
}
template <int BITS, int LAYOUT, int MODE, int MB>
static gemvx_fn pick_gemvx_norm_rd(int rd) {
  switch (rd) {
    case 3012: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 1, 2, 0, false, 3>>;
    case 3022: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2, 0, false, 3>>;
    case 4022: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2, 0, false, 4>>;     // norm + gate / up pair
  }
  return nullptr;
}
template <int BITS, int LAYOUT>
static gemvx_fn pick_gemvx_norm_mode(int mode, int mb, int rd) {
#define WQAA_GXN(MODE) (mb == 1 ? pick_gemvx_norm_rd<BITS, LAYOUT, MODE, 1>(rd) : mb == 2 ? pick_gemvx_norm_rd<BITS, LAYOUT, MODE, 2>(rd) : nullptr)
  switch (mode) {
    case MD_NONE: return WQAA_GXN(MD_NONE);
    case MD_S: return WQAA_GXN(MD_S);
    case MD_ZO: return WQAA_GXN(MD_ZO);
    case MD_ZR: return WQAA_GXN(MD_ZR);
    case MD_ZQ: return WQAA_GXN(MD_ZQ);
  }
#undef WQAA_GXN
  return nullptr;
}
template <int BITS, int LAYOUT, int MODE, int MB>
static gemvx_fn pick_gemvx_rd(int rd) {
  switch (rd) {
    case 12: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 1, 2>>;
    case 22: return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2>>;
    case 13:   // activations in registers
      if constexpr (BITS == 4 && LAYOUT == LAYOUT_LOP3 && MB == 1) return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 1, 2, 0, true>>;
      else return nullptr;
    case 23:
      if constexpr (BITS == 4 && LAYOUT == LAYOUT_LOP3 && MB == 1) return wq_gemvx_kernel<GemvxPolicy<BITS, LAYOUT, MODE, MB, 2, 2, 0, true>>;
      else return nullptr;
  }
  return nullptr;
}
template <int BITS, int LAYOUT>
static gemvx_fn pick_gemvx_mode(int mode, int mb, int rd) {
#define WQAA_GX(MODE) (mb == 1 ? pick_gemvx_rd<BITS, LAYOUT, MODE, 1>(rd) : mb == 2 ? pick_gemvx_rd<BITS, LAYOUT, MODE, 2>(rd) : nullptr)
  switch (mode) {
    case MD_NONE: return WQAA_GX(MD_NONE);
    case MD_S: return WQAA_GX(MD_S);
    case MD_ZO: return WQAA_GX(MD_ZO);
    case MD_ZR: return WQAA_GX(MD_ZR);
    case MD_ZQ: return WQAA_GX(MD_ZQ);
  }
#undef WQAA_GX
  return nullptr;
}
gemvx_fn pick_gemvx_int4(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_int2(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_int1(int layout, int mode, int mb, int rd);
// members with the fused pre/post ops (wqaa_gemvx_inst_pro*.hip)
gemvx_fn pick_gemvx_pro4(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_pro2(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_pro1(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_norm4(int layout, int mode, int mb, int rd);    // RMSNorm in front (wqaa_gemvx_inst_norm*.hip)
gemvx_fn pick_gemvx_norm2(int layout, int mode, int mb, int rd);
gemvx_fn pick_gemvx_norm1(int layout, int mode, int mb, int rd);

}  // namespace wqaa
