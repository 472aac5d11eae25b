// wqaa_chain_kernel.h - ONE persistent launch for a chain of dependent M = 1 GEMVs (wqaa_matmul_chain): the post-attention
// half of a decoder layer, o_proj (+ residual) -> RMSNorm -> gate / up * silu -> down_proj (+ residual), or any chain of
// exact-product GEMV operators whose inputs are earlier operators' outputs.
//
// What it replaces: one launch per operator (reference: bitblas/ops/general_matmul/tilelang/dequantize/
// gemv_dequantize_simt.py:116-262, called once per nn.Linear by integration/BitNet/modeling_bitnet.py:240-244, :839-860).
// On MI355X every dependent launch pays ~2.85 us around a weight stream that itself runs at the memory's rate (DESIGN 3.1:
// dispatch 0.85 + first byte 0.9 + last byte -> last store 1.1), and the WEIGHTS of the next operator do not depend on this
// one's result - only a few KB of activations do.  So (MI355X_MICROARCH.md, rows prefetch-credit / allgather / ldsdma-fill /
// engine-vs-launches):
//   * one workgroup per CU of NL LANES (4), a lane = one LOADER wave + one CONSUMER wave + its slice of an LDS ring.  The
//     loader streams the packed weights of its lane's tasks, operator after operator, into the ring with LDS-DMA
//     (global_load_lds_dwordx4 ... nt: 1 KiB per instruction, no VGPR, no VALU) and never waits on a dependency: when the
//     consumers stall on an edge the rings fill up with the next operator's weights.  Four loaders because ONE wave issues
//     only ~8.4 GB/s of LDS-DMA (tools/dma_lab.hip, profiles/r04_lab_dma_stream.txt: 1 / 2 / 4 loader waves per CU = 2.1 /
//     4.0 / 5.8-6.1 TB/s over the chip, whatever the address pattern, cache policy or queue depth);
//   * the CONSUMER: the exact-product decode / dot of wqaa_gemvx_kernel.h on 16-byte lane chunks read back from the ring
//     (same lane <-> weight bytes map, same per-lane order over the chunks of a row, same wave reduction: a row's bits are
//     the single launch's with kw = 1).  Loaders are waves 0 .. NL-1, consumers NL .. 2NL-1: a workgroup's waves go to the
//     SIMDs round-robin, so every SIMD hosts one loader (SALU + VMEM issue) and one consumer (VALU + LDS);
//   * an operator's output vector crosses to every CU as 8-byte {tag, 2 x float16} granules: one write-through (sc1) store
//     by the lane that holds the two rounded results, swept by ONE consumer wave per CU with relaxed agent-scope loads
//     until every tag matches (cdna_hip_programming.md Guideline 16, recipe R2) and staged into the LDS layout the dots read
//     (the RMSNorm and the residual stash ride in that pass).  No grid barrier, no fence: the data is the flag;
//   * tags are (generation, stage): the generation lives in device memory and is bumped once per launch by workgroup 0,
//     so a replayed hipGraph needs no memset node and stale granules of the previous launch never match.
// Work split: tasks = pairs of output rows (one granule), a contiguous range of tasks per CU, task k of a stage to lane k % NL.  Every spin is bounded (s_memrealtime) and ends in an error code in ctl[1] + abort of the workgroup.
#pragma once
#include "wqaa_gemvx_kernel.h"

namespace wqaa {

constexpr int kChainMaxStages = 8;
constexpr int kChainFill = 4;          // DMA instructions (1 KiB units) per fill
constexpr int kChainLag = 3;           // fills a loader leaves in flight behind its issue point: s_waitcnt vmcnt(12)
constexpr int kChainMaxLanes = 4;      // lanes per workgroup: a loader wave, its slice of the ring, kChainMaxCpl consumer waves at most
constexpr int kChainMaxCpl = 3;        // consumers per lane (tools/chain_task_lab.hip: 4 / 8 / 12 / 16 consumer waves per CU sustain
                                       // 25 / 39 / 49 / 56 GB/s of 4-bit lane chunks - the stream wants 25 and a consumer shares its SIMD)
constexpr int kChainStashMaxRows = 128;

// LDS control block (dwords)
enum : int {
  CL_ABORT = 1,
  CL_GEN = 2,         // this launch's generation, CL_GEN_READY = 1 once valid
  CL_GEN_READY = 3,
  CL_SWEEPING = 4,    // consumers of this CU that are sweeping granules: the loaders thin themselves
  CL_LANDED0 = 8,     // [4] per lane: units of the lane's issue sequence known to have landed
  CL_NEXT0 = 16,      // [4 lanes][4]: ring sequence number (of the lane's unit stream) of each consumer's next unfinished task
  CL_CSTAGE0 = 32,    // [16] stage each consumer has reached (unused slots: INT_MAX)
  CL_SYNC0 = 48,      // [8 stages][4]: arrival counters of the consumers while they stage a stage's input together
  CL_WSUM = 80,       // [16] the norm's per-(virtual)-wave sums of squares
  CL_WORDS = 96
};

// error codes (ctl[1] = code | stage << 8 | wave << 16 | workgroup << 20)
enum : int { CE_LOADER_SPACE = 1, CE_LOADER_SC = 2, CE_WAIT_LANDED = 3, CE_WAIT_ACT = 4, CE_SWEEP = 5, CE_WAIT_GEN = 6, CE_WAIT_STAGE = 7 };

struct ChainStage {
  const void* A;            // in_kind 0: (K,) float16
  const void* B[2];         // [1]: the `up` operator of a gate / up pair
  const void* scale[2];
  const void* zeros[2];
  const void* bias[2];
  const void* residual;     // (N,) float16 from the caller, or NULL
  const void* norm_weight;  // (K,) float16: RMSNorm in front, or NULL
  void* C;                  // (N,) float16, or NULL (only later stages read it)
  float norm_eps, norm_inv_k;
  int N, K, kg, gq_shift;
  uint32_t gq_magic;
  int nc, cpr, row_bytes;
  int zint;
  uint32_t flip;
  int has_bias;
  int in_kind;              // 0 caller's A, 1 granules of stage `src`, 2 the staged input of the previous stage
  int src;
  int pair;
  int publish;              // granules: a later stage reads this output
  int res_stage;            // -1, or j: residual = output of stage j, stashed while a stage up to this one swept it
  int stash_for;            // -1, or s2: while sweeping my input keep the rows stage s2 of this CU adds as its residual
  int norm_nwv, norm_nai;   // the single launch's geometry (waves, items per thread): the order of its sum of squares
  int gran_off;             // output granules in the workspace
  int tasks;                // ceil(N / 2)
  int un;                   // ring units per task: (pair ? 4 : 2) * nc
  int sc_units;             // 1 KiB units per (operator, scale | zeros) block
  int nsc;                  // scale / zeros units at the head of the stage's stream
  int wait_stage;           // the stager waits until every consumer has reached this stage (its LDS input buffer is free)
  int a_off, sa_off, sc_off, stash_off;   // LDS byte offsets
};

struct ChainArgs {
  ChainStage st[kChainMaxStages];
  int nstages;
  int nlanes, cpl;          // lanes and consumers per lane (blockDim = 64 * nlanes * (1 + cpl))
  int g_shift;              // log2(grid) when the grid is a power of two (256 CUs), else -1
  int ring_off, ring_units; // ring_units: per lane; lane l's slice starts at ring_off + l * ring_units KiB
  int parts_off;
  int bump_stage;           // the stage after whose sweep workgroup 0 bumps the generation (-1: no edge)
  int thin;                 // 1: one fill outstanding while a consumer of this CU sweeps
  int sweep_sleep;          // naps of ~0.2 us between two reads of an incomplete sweep
  unsigned timeout_ticks;   // s_memrealtime ticks (100 MHz) a wait may take
  int lab;                  // tools only: 1 no dots, 2 consumers do not wait for the weights, 4 no sweeps (tiles "ready" at once), 8 default-policy DMA, 16 no weight stream, 32 consumers at s_setprio 3 (1 - 16: results wrong by construction)
  unsigned long long* gran;
  uint32_t* ctl;            // [0] generation, [1] first error
  unsigned long long* trace;  // lab: [workgroup][wave][32] time stamps, or NULL
};

typedef __attribute__((address_space(1))) unsigned long long chain_gu64;
typedef __attribute__((address_space(1))) unsigned int chain_gu32;
// every global operand the consumers touch goes through a GLOBAL-address-space pointer: a generic pointer makes a FLAT
// instruction, which also counts on lgkmcnt - an LDS wait would then sit out a memory round trip
#define CHAIN_G(T, p) (reinterpret_cast<const __attribute__((address_space(1))) T*>((const __attribute__((address_space(1))) void*)(p)))
#define CHAIN_GW(T, p) (reinterpret_cast<__attribute__((address_space(1))) T*>((__attribute__((address_space(1))) void*)(p)))

// control words: volatile accesses through LDS-address-space pointers (a volatile access through a generic pointer is a FLAT
// instruction, which counts on vmcnt - the loader's DMA counter)
typedef __attribute__((address_space(3))) uint32_t chain_lds_u32;
typedef __attribute__((address_space(3))) unsigned char chain_lds_u8;
typedef __attribute__((address_space(3))) u32x4 chain_lds_u32x4;
// (every lane reads the same word: handed back as a SCALAR - control flow that hangs on a vector value is compiled into
// exec-masked regions, and every wait loop of this kernel hangs on one of these)
__device__ __forceinline__ uint32_t chain_lds_ld(const unsigned char* smem, int word) {
  return __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile chain_lds_u32*>((const chain_lds_u8*)smem + word * 4));
}
__device__ __forceinline__ void chain_lds_st(unsigned char* smem, int word, uint32_t v) {
  *reinterpret_cast<volatile chain_lds_u32*>((chain_lds_u8*)smem + word * 4) = v;
}
// min of the four / sixteen control words at `word` (one value per consumer slot; unused slots hold INT_MAX)
__device__ __forceinline__ int chain_lds_min4(const unsigned char* smem, int word) {
  const u32x4 a = *reinterpret_cast<const volatile chain_lds_u32x4*>((const chain_lds_u8*)smem + word * 4);
  int m = (int)a[0];
  m = (int)a[1] < m ? (int)a[1] : m;
  m = (int)a[2] < m ? (int)a[2] : m;
  m = (int)a[3] < m ? (int)a[3] : m;
  return __builtin_amdgcn_readfirstlane(m);
}
__device__ __forceinline__ int chain_lds_min16(const unsigned char* smem, int word) {
  const int a = chain_lds_min4(smem, word), b = chain_lds_min4(smem, word + 4), c = chain_lds_min4(smem, word + 8), d = chain_lds_min4(smem, word + 12);
  const int m = a < b ? a : b, n = c < d ? c : d;
  return m < n ? m : n;
}
// order this wave's LDS accesses against a control word (LDS only: a workgroup fence over every address space would wait
// for the wave's outstanding global stores, a memory round trip per task)
#define CHAIN_LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define CHAIN_LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")

// one 1 KiB LDS-DMA unit: lane l's 16 bytes at sbase + voff land at LDS byte lds_dst + 16 * l.  M0 is saved and restored
// (the compiler owns it); the s_nop 4 covers v_readfirstlane -> SGPR -> VMEM base, the s_nop 0 M0 -> LDS-DMA.
__device__ __forceinline__ void chain_dma(unsigned lds_dst, unsigned voff, unsigned long long sbase, bool default_policy = false) {
  unsigned keep;
  if (default_policy) {      // (lab: the weight stream without the non-temporal hint)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(sbase)
                 : "memory");
    return;
  }
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_dst), "s"(sbase)
               : "memory");
}

__device__ __forceinline__ unsigned long long chain_uniform64(unsigned long long x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

struct ChainWave {
  unsigned char* smem;
  const ChainArgs* args;
  int lane, wave, b, G;
  unsigned timeout;
  __device__ __forceinline__ void stamp(int i) const {
    if (args->trace && lane == 0) {
      args->trace[((long)b * 16 + wave) * 64 + i] = __builtin_amdgcn_s_memrealtime();
      if (i == 0 || i == 3) args->trace[((long)b * 16 + wave) * 64 + 24 + (i ? 1 : 0)] = __builtin_amdgcn_s_memtime();      // shader clock at wave start / end
    }
  }
  __device__ __forceinline__ void fail(int code, int stage) const {
    if (lane == 0) {
      unsigned expected = 0;
      __hip_atomic_compare_exchange_strong((chain_gu32*)(args->ctl + 1), &expected, (unsigned)(code | (stage << 8) | (wave << 16) | (b << 20)),
                                           __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    chain_lds_st(smem, CL_ABORT, 1u);
  }
  // every 32 polls: give up when the workgroup has aborted or the wait has lasted longer than the limit
  __device__ __forceinline__ bool expired(unsigned& n, unsigned long long& t0) const {
    __builtin_amdgcn_s_sleep(1);
    if ((++n & 31u) != 0) return false;
    if (chain_lds_ld(smem, CL_ABORT) != 0) return true;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) {
      t0 = now;
      return false;
    }
    return now - t0 > (unsigned long long)timeout;
  }
  // LDS word >= value
  __device__ __forceinline__ bool wait_ge(int word, uint32_t value, int code, int stage) const {
    unsigned n = 0;
    unsigned long long t0 = 0;
    while ((int)chain_lds_ld(smem, word) < (int)value) {
      if (expired(n, t0)) {
        fail(code, stage);
        return false;
      }
    }
    return true;
  }
  __device__ __forceinline__ bool wait_cstage(int stage, int code, int at) const {
    unsigned n = 0;
    unsigned long long t0 = 0;
    while (chain_lds_min16(smem, CL_CSTAGE0) < stage) {
      if (expired(n, t0)) {
        fail(code, at);
        return false;
      }
    }
    return true;
  }
  __device__ __forceinline__ void task_range(int tasks, int& t0, int& t1) const {
    const int gs = args->g_shift;
    if (gs >= 0) {                                                        // tasks * G < 2^32 (host)
      t0 = (int)(((uint32_t)b * (uint32_t)tasks) >> gs);
      t1 = (int)(((uint32_t)(b + 1) * (uint32_t)tasks) >> gs);
    } else {
      t0 = (int)(((uint32_t)b * (uint32_t)tasks) / (uint32_t)G);
      t1 = (int)(((uint32_t)(b + 1) * (uint32_t)tasks) / (uint32_t)G);
    }
  }
};

// ---- a loader wave (lane L of NL) -------------------------------------------------------------------------------------------
template <class P>
__device__ void chain_loader(const ChainWave& cw) {
  const ChainArgs& args = *cw.args;
  unsigned char* smem = cw.smem;
  const int lane = cw.lane;
  const int L = cw.wave, NL = args.nlanes;
  int RING = args.ring_units, thin = args.thin;
  int ring_base = args.ring_off + L * RING * 1024;
  asm volatile("" : "+s"(RING), "+s"(ring_base), "+s"(thin));
  const unsigned ring_end = (unsigned)(ring_base + RING * 1024);
  int issued = 0;         // units this loader has issued (lane 0: the scale blocks too)
  int rseq = 0;           // ring units this loader has issued
  unsigned dst = (unsigned)ring_base;   // LDS byte address of ring slot rseq % RING
  int in_fill = 0;
  int frontier = 0;       // cached: min over the lane's consumers of the ring sequence number of their next unfinished task
  int landed_pub = 0;
  bool dead = false;
  const bool dflt = (args.lab & 8) != 0;
  const unsigned voff_full = (unsigned)lane * 16u;
  cw.stamp(1);
  if (args.lab & 16) {          // lab: no weight stream at all - the consumers run on whatever the ring holds
    chain_lds_st(smem, CL_LANDED0 + L, 0x7ffffff0u);
    return;
  }

  auto publish = [&](int landed) {
    if (landed > landed_pub) {
      landed_pub = landed;
      chain_lds_st(smem, CL_LANDED0 + L, (uint32_t)landed);
    }
  };
  auto has_space = [&]() { return rseq + kChainFill <= frontier + RING; };
  // the ring slice is full: publish what is in flight as it lands (the consumer may be waiting for exactly that), fill by
  // fill, and go on as soon as a fill's worth of slots is free
  auto wait_space = [&](int stage) {
#define CHAIN_DRAIN_STEP(N)                                   \
  asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");       \
  publish(issued - N);                                        \
  frontier = chain_lds_min4(smem, CL_NEXT0 + 4 * L);          \
  if (has_space()) return;
    CHAIN_DRAIN_STEP(8)
    CHAIN_DRAIN_STEP(4)
    CHAIN_DRAIN_STEP(0)
#undef CHAIN_DRAIN_STEP
    unsigned n_ = 0;
    unsigned long long t_ = 0;
    for (;;) {
      frontier = chain_lds_min4(smem, CL_NEXT0 + 4 * L);
      if (has_space()) return;
      if (cw.expired(n_, t_)) {
        cw.fail(CE_LOADER_SPACE, stage);
        dead = true;
        return;
      }
    }
  };
  // a fill is complete: leave kChainLag fills (one while a consumer of this CU sweeps) in flight, publish the rest, and make
  // sure the next fill's ring slots are free
  auto boundary = [&](int stage) {
    in_fill = 0;
    if (thin && chain_lds_ld(smem, CL_SWEEPING) != 0) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      publish(issued - kChainFill);
    } else {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      publish(issued - kChainFill * kChainLag);
    }
    static_assert(kChainFill == 4 && kChainLag == 3, "the vmcnt immediates above");
    if (!has_space()) {
      frontier = chain_lds_min4(smem, CL_NEXT0 + 4 * L);
      if (!has_space()) wait_space(stage);
    }
  };

  for (int s = 0; s < args.nstages && !dead; ++s) {
    const ChainStage& S = args.st[s];
    int t0, t1;
    cw.task_range(S.tasks, t0, t1);
    const int nt = t1 - t0;
    const int n0 = 2 * t0;
    const int nops = S.pair ? 2 : 1;
    // ---- scale / zeros blocks of this CU's rows (lane 0): contiguous in the (N, K / g) tensors, 16-byte windows aligned in
    // absolute address (a window never straddles a page), lanes past the block re-read its first window ----
    if (L == 0 && S.nsc > 0) {
      if (s >= 2 && chain_lds_min16(smem, CL_CSTAGE0) < s - 1) {     // the block of stage s - 2 lives in the same LDS area
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(issued);
        if (!cw.wait_cstage(s - 1, CE_LOADER_SC, s)) return;
      }
      int nr = 2 * nt;
      if (n0 + nr > S.N) nr = S.N - n0;
      if (nr < 0) nr = 0;
      int blk = 0;
      constexpr int NTENS = (P::MODE == MD_ZO || P::MODE == MD_ZR) ? 2 : 1;
      for (int op = 0; op < nops; ++op) {
#pragma unroll
        for (int tn = 0; tn < NTENS; ++tn) {
          const unsigned char* base = reinterpret_cast<const unsigned char*>(tn == 0 ? S.scale[op] : S.zeros[op]);
          const unsigned long long beg = (unsigned long long)(base) + (unsigned long long)((long)n0 * S.kg * 2);
          const unsigned long long a16 = chain_uniform64(beg & ~15ull);
          const unsigned span = (unsigned)(beg - a16) + (unsigned)nr * (unsigned)S.kg * 2u;     // bytes from a16 to the block's end
          for (int u = 0; u < S.sc_units; ++u) {
            unsigned off = (unsigned)u * 1024u + voff_full;
            off = off < span ? off : 0u;
            chain_dma((unsigned)(S.sc_off + (blk * S.sc_units + u) * 1024), off, a16);
            ++issued;
            if (++in_fill == kChainFill) {
              boundary(s);
              if (dead) return;
            }
          }
          ++blk;
        }
      }
    }
    // ---- the weight rows of this lane's tasks (t0 + L, t0 + L + NL, ...), in the order its consumer reads them back ----
    int nc = S.nc, rows_per_task = S.pair ? 4 : 2, pair = S.pair, Nrows = S.N, row_bytes = S.row_bytes;
    unsigned long long B0 = (unsigned long long)S.B[0], B1 = (unsigned long long)S.B[S.pair ? 1 : 0];
    const bool tail_partial = (S.cpr & 63) != 0;
    unsigned voff_tail = voff_full;
    {
      const int chunk = (nc - 1) * 64 + lane;
      if (chunk >= S.cpr) voff_tail = 0u;        // lanes past the row re-read the unit's first 16 bytes: they meet zero activations
    }
    int nfull = tail_partial ? nc - 1 : nc;
    asm volatile("" : "+s"(nc), "+s"(rows_per_task), "+s"(pair), "+s"(Nrows), "+s"(row_bytes), "+s"(B0), "+s"(B1), "+s"(nfull));
    for (int t = t0 + L; t < t1; t += NL) {
      // the task's rows, lane chunk by lane chunk (chunk c of every row, then chunk c + 1): its consumer starts on chunk 0 while
      // the rest is still in flight, and hands the ring slots back chunk by chunk
      unsigned long long rowb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = 2 * t + (pair ? (r >> 1) : (r & 1));
        n = n < Nrows ? n : Nrows - 1;
        const unsigned long long base = (pair && (r & 1)) ? B1 : B0;
        rowb[r] = chain_uniform64(base + (unsigned long long)((long)n * row_bytes));
      }
      for (int c = 0; c < nc; ++c) {
        const unsigned voff = c < nfull ? voff_full : voff_tail;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r >= rows_per_task) break;
          chain_dma(dst, voff, rowb[r] + (unsigned long long)c * 1024ull, dflt);
          dst += 1024u;
          if (dst == ring_end) dst = (unsigned)ring_base;
          ++rseq;
          ++issued;
          if (++in_fill == kChainFill) {
            if (L == 0 && (issued & 15) == 0 && issued <= 64) cw.stamp(19 + (issued >> 4));      // lab: lane 0 has issued 16 / 32 / 48 / 64 units
            boundary(s);
            if (dead) return;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  publish(issued);
  cw.stamp(2);
}

// 4-bit LOP3 weights (`fast_decoding`, the default of W_int4 x A_fp16): the interleave puts consecutive elements 2j, 2j + 1
// into the two halves of field j, so the order the unpack produces IS memory order (KindTraits::src_elem is the identity).
// The chain then keeps an operator's input in LDS as it is: nothing to permute when it arrives, and the chunk sums are
// taken by the consumer from the activations it loads anyway.
template <class P>
constexpr bool chain_natural_layout() {
  for (int x = 0; x < P::EPW; ++x)
    if (P::T::src_elem(P::LAYOUT, x) != x) return false;
  return true;
}
template <class P>
constexpr bool kChainNatural = chain_natural_layout<P>();

// ---- one lane chunk of ROWS weight rows against the staged activations: per row the arithmetic of wq_gemvx_kernel's
// `consume`, MB = 1, to the letter (class accumulators over the four words, Horner, zero point through the chunk's
// activation sum, group scale on the fp32 partial).  The rows advance together, innermost: one consumer wave per SIMD has no
// other wave to cover the latency of a dependent V_DOT2C chain, the 2 x ROWS independent chains of a task do ----
template <class P, int ROWS>
__device__ __forceinline__ void chain_chunk(const u32x4 (&w)[ROWS], const u32x4 (&av)[4 * P::PPW], const float sa, const uint32_t (&sbits)[ROWS],
                                            const uint32_t (&zbits)[ROWS], const float zint, const uint32_t flip, float (&acc)[ROWS]) {
  constexpr int BITS = P::BITS, NPAIR = P::NPAIR, NCLS = P::NCLS, PPW = P::PPW, MODE = P::MODE;
  float cls[ROWS][NCLS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int k = 0; k < NCLS; ++k) cls[r][k] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    uint32_t f[ROWS][NPAIR];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const uint32_t ww = BITS == 1 ? (w[r][u] ^ flip) : w[r][u];
      const uint32_t w8 = ww >> 8;
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        constexpr uint32_t fmask = ((1u << BITS) - 1u) * 0x00010001u;
        const int bit = BITS * i;
        f[r][i] = (bit >= 8 ? w8 : ww) & (fmask << (bit & 7));
      }
    }
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
      const u32x4 a = av[u * PPW + pp];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = pp * 4 + e;
        const int k = ((BITS * i) & 7) / BITS;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) cls[r][k] = __builtin_amdgcn_fdot2(as_h2(f[r][i]), as_h2(a[e]), cls[r][k], false);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float t = cls[r][NCLS - 1];
#pragma unroll
    for (int k = NCLS - 2; k >= 0; --k) t = __builtin_fmaf(t, 1.f / (float)(1 << BITS), cls[r][k]);
    t *= 16777216.f;
    float z = zint;
    if constexpr (MODE == MD_ZO) z += (float)bits_to_half(zbits[r]);
    t = __builtin_fmaf(-z, sa, t);
    if constexpr (MODE == MD_NONE) {
      acc[r] += t;
    } else {
      acc[r] = __builtin_fmaf(t, (float)bits_to_half(sbits[r]), acc[r]);
      if constexpr (MODE == MD_ZR) acc[r] = __builtin_fmaf(-(float)bits_to_half(zbits[r]), sa, acc[r]);
    }
  }
}

// ---- staging: EPW activations (one item = the partner of one weight word of one lane chunk) -> the LDS tile in the order
// the unpack produces + the chunk's activation sum; wq_gemvx_kernel's `item_store`, with the four partial sums of a lane
// chunk combined here ((p0 + p1) + (p2 + p3), the order `consume` adds them in) by the four lanes that hold them ----
template <class P>
__device__ __forceinline__ void chain_item_store(unsigned char* smem, int a_off, int sa_off, int c, int u, int l, const u32x4 (&raw)[P::EPW / 8],
                                                 bool valid, int lane) {
  using T = typename P::T;
  constexpr int EPW = P::EPW, PPW = P::PPW, PIECES = P::PIECES;
  u32x4* a_lds = reinterpret_cast<u32x4*>(smem + a_off);
  float* sa_lds = reinterpret_cast<float*>(smem + sa_off);
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
    a_lds[((long)c * PIECES + u * PPW + pp) * 64 + l] = out;
  }
  // lanes 4j .. 4j + 3 hold the partials u = 0 .. 3 of one lane chunk (the callers' item order)
  float p = valid ? sum : 0.f;
  const float q = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  p = p + q;                                                                                                                       // u even: p_u + p_(u+1)
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  p = p + r2;                                                                                                                      // lane u = 0: (p0 + p1) + (p2 + p3)
  if ((lane & 3) == 0) sa_lds[c * 64 + l] = p;
}

// stage lane chunk c IN PLACE: its 64 * E activations sit in the chunk's own region of the tile in natural memory order
// (item t of the chunk = EPW elements at byte t * EPW * 2; the permuted tile of a chunk fills exactly the same bytes); this wave
// reads all four of its items per lane, then writes them back in the order the unpack produces (+ the chunk's activation sums).
// One wave per chunk: its LDS reads are in the queue ahead of its writes.  NORM: x -> weight * half(x * r) first (the norm's two
// roundings), weight = nw[q] of item q * 64 + lane of the chunk.
template <class P, bool NORM>
__device__ __forceinline__ void chain_stage_chunk(unsigned char* smem, int a_off, int sa_off, int cpr, int c, int lane, float norm_r,
                                                  const u32x4 (*nw)[P::EPW / 8]) {
  constexpr int EPW = P::EPW, IVW = EPW / 8, E = P::E;
  const unsigned char* src = smem + a_off + (long)c * (64 * E * 2);
  u32x4 raw[4][IVW];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int v = 0; v < IVW; ++v) raw[q][v] = reinterpret_cast<const u32x4*>(src + (long)(q * 64 + lane) * (EPW * 2))[v];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = q * 64 + lane;               // item of the chunk: lane chunk position t >> 2, weight word t & 3
    const int l = t >> 2, u = t & 3;
    const bool valid = c * 64 + l < cpr;
    if constexpr (NORM) {
#pragma unroll
      for (int v = 0; v < IVW; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t x = as_h2(raw[q][v][e]);
          const half2_t h = {(half_t)((float)x[0] * norm_r), (half_t)((float)x[1] * norm_r)};
          raw[q][v][e] = as_u32(as_h2(nw[q][v][e]) * h);
        }
    }
    chain_item_store<P>(smem, a_off, sa_off, c, u, l, raw[q], valid, lane);
  }
}

// sum x^2 of the raw vector (natural order, in the tile's region) in the order the single launch takes it (NWV waves x NAI
// items per thread: item idx = j * threads + tid; per thread over j, per wave by the DPP ladder, across the waves in wave
// order).  (1) the consumer that brought lane chunk c in takes the per-item partial sums of the chunk's four slots
// (launch-path item idx = c * 256 + u * 64 + l, slot = idx >> 6 = 4c + u); (2) after ONE meeting every staging consumer
// runs the wave ladders and the sum over the waves itself - the same bits in every wave, no second hand-off.
template <class P>
__device__ __forceinline__ void chain_norm_parts(unsigned char* smem, const ChainArgs& args, const ChainStage& S, int lane, int c) {
  constexpr int EPW = P::EPW, IVW = EPW / 8;
  float* parts = reinterpret_cast<float*>(smem + args.parts_off);      // [slot][lane]
  const unsigned char* raw = smem + S.a_off;
  const int cl = c * 64 + lane;
  const bool valid = cl < S.cpr;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float part = 0.f;
    if (valid) {
      const u32x4* src = reinterpret_cast<const u32x4*>(raw + ((long)cl * 4 + u) * (EPW * 2));
#pragma unroll
      for (int v = 0; v < IVW; ++v) {
        const u32x4 x = src[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) part = __builtin_amdgcn_fdot2(as_h2(x[e]), as_h2(x[e]), part, false);
      }
    }
    parts[(c * 4 + u) * 64 + lane] = part;
  }
}
__device__ __forceinline__ float chain_norm_rinv(const unsigned char* smem, const ChainArgs& args, const ChainStage& S, int lane) {
  const float* parts = reinterpret_cast<const float*>(smem + args.parts_off);
  const int nslots = S.nc * 4;
  const int nwv = S.norm_nwv, nai = S.norm_nai;
  float tot = 0.f;
  for (int w = 0; w < nwv; ++w) {
    float ssq = 0.f;
    for (int j = 0; j < nai; ++j) {
      const int sl = j * nwv + w;
      ssq += sl < nslots ? parts[sl * 64 + lane] : 0.f;
    }
    const float ws = wave_sum_l63(ssq);
    const float wsum = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ws), 63));
    tot = w == 0 ? wsum : tot + wsum;
  }
  return rsqrtf(tot * S.norm_inv_k + S.norm_eps);
}

// ---- the consumer waves ---------------------------------------------------------------------------------------------------
// what a stage's tasks need, fetched from the kernel-argument segment once per stage
struct ChainTaskCtx {
  int nc, cpr, kg, gq_shift, N, n0;
  uint32_t gq_magic, flip;
  int a_off, sa_off, ring_off, ring_units;
  int sc_rel[2], z_rel[2];      // LDS byte address of (row n0, group 0) in the scale / zeros block of operator 0 / 1
  float zint;
  int has_bias, has_res, stash_off;
  const void* bias[2];
  const void* residual;
  void* C;
  unsigned long long* gran;     // this stage's granules, or NULL
  uint32_t tag;
};

// one task: ROWS weight rows (two output elements) against the staged input.  The rows sit in the lane's ring lane chunk by
// lane chunk (slot rpos + c * ROWS + r); chunk c is awaited on its own (the loader's landed count `need0 + (c + 1) * ROWS`) and
// its slots are handed back as soon as its dots are done (`next_word` = the ring sequence number of this consumer's first
// unit still in use).  false: the wait gave up.
template <class P, int ROWS>
__device__ __forceinline__ bool chain_task(const ChainWave& cw, const ChainTaskCtx& X, int t, int rpos, int landed_word, int need0, int next_word,
                                           int rseq0, int stage, bool free_run) {
  constexpr int PPW = P::PPW, PIECES = P::PIECES, MODE = P::MODE;
  constexpr bool PAIR = ROWS == 4;
  constexpr bool ZT = MODE == MD_ZO || MODE == MD_ZR;
  unsigned char* smem = cw.smem;
  const int lane = cw.lane;
  const int RING = X.ring_units;
  const u32x4* a_lds = reinterpret_cast<const u32x4*>(smem + X.a_off);
  const float* sa_lds = reinterpret_cast<const float*>(smem + X.sa_off);
  // output elements of the task, the operator and scale block of each streamed row
  int elem[ROWS], sc_base[ROWS], z_base[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    int e = 2 * t + (PAIR ? (r >> 1) : r);
    e = e < X.N ? e : X.N - 1;
    elem[r] = e;
    const int op = PAIR ? (r & 1) : 0;
    sc_base[r] = X.sc_rel[op] + (e - X.n0) * X.kg * 2;
    z_base[r] = X.z_rel[op] + (e - X.n0) * X.kg * 2;
  }
  // bias / the caller's residual: asked for before the dots (every lane the same address: one request), used behind them
  uint16_t res_bits[ROWS], bias_bits[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    res_bits[r] = 0;
    bias_bits[r] = 0;
    if (X.has_bias) bias_bits[r] = CHAIN_G(uint16_t, X.bias[PAIR ? (r & 1) : 0])[elem[r]];
    if constexpr (!PAIR) {
      if (X.residual) res_bits[r] = CHAIN_G(uint16_t, X.residual)[elem[r]];
    }
  }
  struct Ops {
    u32x4 av[4 * PPW];
    float sa;
    u32x4 w[ROWS];
    uint32_t sb[ROWS], zb[ROWS];
  };
  int slot = rpos;                                  // ring slot of (chunk c, row 0), c = the next chunk to load
  int landed = (int)chain_lds_ld(smem, landed_word);
  // every LDS operand of lane chunk c, asked for in one go, once the loader says the chunk has landed
  auto load = [&](Ops& o, int c) -> bool {
    const int need = need0 + (c + 1) * ROWS;
    if (!free_run && landed < need) {
      unsigned n_ = 0;
      unsigned long long t_ = 0;
      for (;;) {
        landed = (int)chain_lds_ld(smem, landed_word);
        if (landed >= need) break;
        if (cw.expired(n_, t_)) {
          cw.fail(CE_WAIT_LANDED, stage);
          return false;
        }
      }
      CHAIN_LDS_ACQUIRE();
    }
    const int chunk = c * 64 + lane;
    int gi = 0;
    if constexpr (MODE != MD_NONE) {
      const int ch = chunk < X.cpr ? chunk : 0;           // (the single launch clamps the chunk before it takes the group)
      gi = X.gq_shift >= 0 ? (ch >> X.gq_shift) : (int)__umulhi((uint32_t)ch, X.gq_magic);
    }
    if constexpr (kChainNatural<P>) {
      // the tile is the input vector itself (natural order: the unpack's order IS memory order for this layout): the lane's
      // four items are 64 contiguous bytes; the chunk's activation sum is taken here, by the additions the staged tile's is
      // (one partial per weight word, then (p0 + p1) + (p2 + p3))
#pragma unroll
      for (int j = 0; j < 4; ++j) o.av[j] = a_lds[(long)chunk * 4 + j];
      float part[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        part[u] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[u] = __builtin_amdgcn_fdot2(as_h2(o.av[u][e]), half2_t{(half_t)1.f, (half_t)1.f}, part[u], false);
      }
      o.sa = (part[0] + part[1]) + (part[2] + part[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4 * PPW; ++j) o.av[j] = a_lds[((long)c * PIECES + j) * 64 + lane];
      o.sa = sa_lds[c * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      o.w[r] = *reinterpret_cast<const u32x4*>(smem + X.ring_off + slot * 1024 + lane * 16);
      if (++slot == RING) slot = 0;
      o.sb[r] = 0;
      o.zb[r] = 0;
      if constexpr (MODE != MD_NONE) o.sb[r] = *reinterpret_cast<const uint16_t*>(smem + sc_base[r] + gi * 2);
      if constexpr (ZT) o.zb[r] = *reinterpret_cast<const uint16_t*>(smem + z_base[r] + gi * 2);
    }
    return true;
  };
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  // the dots of chunk c; its ring slots go back to the loader (the operands are in registers)
  auto compute = [&](const Ops& o, int c) {
    chain_lds_st(smem, next_word, (uint32_t)(rseq0 + (c + 1) * ROWS));
    chain_chunk<P, ROWS>(o.w, o.av, o.sa, o.sb, o.zb, X.zint, X.flip, acc);
  };
  if constexpr (ROWS <= 2) {
    // two rows: the next chunk's operands are in flight while this one's dots run
    Ops oa, ob;
    const int nc = X.nc;
    if (!load(oa, 0)) return false;
    int c = 0;
    for (;;) {
      if (c + 1 < nc && !load(ob, c + 1)) return false;
      __builtin_amdgcn_sched_barrier(0);
      compute(oa, c);
      if (++c >= nc) break;
      if (c + 1 < nc && !load(oa, c + 1)) return false;
      __builtin_amdgcn_sched_barrier(0);
      compute(ob, c);
      if (++c >= nc) break;
    }
  } else {
    // four rows (a gate / up task): one set of operands (the kernel runs four waves per SIMD in 128 registers: the other
    // waves cover the LDS round trip)
    Ops oa;
    for (int c = 0; c < X.nc; ++c) {
      if (!load(oa, c)) return false;
      __builtin_amdgcn_sched_barrier(0);
      compute(oa, c);
    }
  }
  float tot[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) tot[r] = wave_sum_l63(acc[r]);
  if (lane == 63) {
    half_t out[2];
    const int e0 = 2 * t;
    const bool two = e0 + 1 < X.N;
    if constexpr (PAIR) {
      // both projections' results (+ their biases) rounded to float16 as their own launches would store them, then
      // torch's F.silu(gate) * up (wq_gemvx_kernel, PAIR members)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half_t h[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          h[q] = (half_t)tot[2 * j + q];
          if (X.has_bias) h[q] = h[q] + bits_to_half(bias_bits[2 * j + q]);
        }
        out[j] = silu_mul_h(h[0], h[1]);
      }
    } else {
      // result (+ bias) rounded to float16, then + residual in fp32, rounded again (wq_gemvx_kernel, PRO members)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        half_t h = (half_t)tot[r];
        if (X.has_bias) h = h + bits_to_half(bias_bits[r]);
        if (X.has_res) {
          const half_t rv = X.residual ? bits_to_half(res_bits[r]) : reinterpret_cast<const half_t*>(smem + X.stash_off)[elem[r] - X.n0];
          h = (half_t)((float)h + (float)rv);
        }
        out[r] = h;
      }
    }
    if (!two) out[1] = (half_t)0.f;
    const uint32_t bits = as_u32(half2_t{out[0], out[1]});
    if (X.C) {
      if (two) CHAIN_GW(uint32_t, X.C)[t] = bits;                 // elements 2t, 2t + 1: one aligned 4-byte store
      else CHAIN_GW(uint16_t, X.C)[e0] = (uint16_t)(bits & 0xFFFFu);
    }
    if (X.gran) __hip_atomic_store((chain_gu64*)(X.gran + t), ((unsigned long long)X.tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}

template <class P>
__device__ void chain_consumer(const ChainWave& cw) {
  constexpr int EPW = P::EPW, IVW = EPW / 8, E = P::E, MODE = P::MODE;
  constexpr int PPC = E / 32;                     // sweep passes (1024 granules = 2048 elements) per lane chunk
  constexpr int NTENS = (MODE == MD_ZO || MODE == MD_ZR) ? 2 : 1;
  const ChainArgs& args = *cw.args;
  unsigned char* smem = cw.smem;
  const int lane = cw.lane;
  const int NL = args.nlanes, CPL = args.cpl, NCONS = NL * CPL;
  const int ci = cw.wave - NL;                    // consumer index: lane ci % NL (on that lane's loader's SIMD), position ci / NL in it
  const int ln = ci % NL, sub = ci / NL;
  const int RING = args.ring_units;
  int rseq_base = 0, useq_base = 0;               // of the current stage, in this lane's own unit stream
  int useq0_base = 0;                             // ... in lane 0's (which also carries every stage's scale blocks)
  // lab: shader cycles this wave spent waiting for weights / in the dots / staging inputs / in the staging's meetings / waiting for granules
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
  const bool timing = args.trace != nullptr;
  auto now = [&]() -> unsigned long long { return timing ? __builtin_amdgcn_s_memtime() : 0ull; };
  // lab: shader-clock marks of this wave, ten per stage (slots 32 + 10 * stage + i), first three stages
  auto cyc = [&](int stage, int i) {
    if (timing && stage < 3 && lane == 0) args.trace[((long)cw.b * 16 + cw.wave) * 64 + 32 + 10 * stage + i] = __builtin_amdgcn_s_memtime();
  };
  uint32_t gen = 0;
  bool have_gen = false;
  auto need_gen = [&](int s) -> bool {
    if (have_gen) return true;
    if (!cw.wait_ge(CL_GEN_READY, 1u, CE_WAIT_GEN, s)) return false;
    gen = chain_lds_ld(smem, CL_GEN);
    have_gen = true;
    return true;
  };
  if (ci == 0) {
    // the workgroup's generation: ONE agent-scope load, shared through LDS (every tag of this launch derives from it)
    const uint32_t g = __hip_atomic_load((chain_gu32*)args.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_lds_st(smem, CL_GEN, g);
    CHAIN_LDS_RELEASE();
    chain_lds_st(smem, CL_GEN_READY, 1u);
    gen = g;
    have_gen = true;
  }

  for (int s = 0; s < args.nstages; ++s) {
    const ChainStage& S = args.st[s];
    int t0, t1;
    cw.task_range(S.tasks, t0, t1);
    const int nt = t1 - t0;
    const int n0 = 2 * t0;
    chain_lds_st(smem, CL_CSTAGE0 + ci, (uint32_t)s);
    cyc(s, 9);
    // ---- the stage's input: staged once per CU by its consumers TOGETHER, lane chunk c (64 * E elements) by consumer c % NCONS,
    // in place in the chunk's region of the tile ----
    if (S.in_kind != 2) {
      cw.stamp(4 + 3 * s);
      const unsigned long long tst0 = now();
      // the LDS tile of this input generation was read by the stages two generations back
      if (S.wait_stage > 0 && !cw.wait_cstage(S.wait_stage, CE_WAIT_STAGE, s)) return;
      const int a_off = S.a_off, sa_off = S.sa_off, cpr = S.cpr, nc = S.nc;
      const bool norm = S.norm_weight != nullptr;
      auto mark = [&](int i) { cyc(s, i); };
      mark(0);
      auto sync = [&](int which) -> bool {
        const unsigned long long ts0 = now();
        CHAIN_LDS_RELEASE();
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(smem) + CL_SYNC0 + s * 4 + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!cw.wait_ge(CL_SYNC0 + s * 4 + which, (uint32_t)NCONS, CE_WAIT_ACT, s)) return false;
        CHAIN_LDS_ACQUIRE();
        acc_t[3] += now() - ts0;
        return true;
      };
      if constexpr (kChainNatural<P>) {
        // ---- natural-order tile: a unit of work is one WEIGHT-WORD SLOT sl = 4c + u of the vector - per lane l the item
        // (8 activations, 16 bytes, four granules) of lane chunk position c * 64 + l, word u: exactly the items ONE thread
        // of the single launch loads (item idx = sl * 64 + l).  Without a norm the items go straight to the tile.  With one,
        // a consumer takes every slot of ONE (virtual) wave w of the launch (slots w, w + nwv, ...): it owns that wave's
        // whole sum of squares - per thread over its items, the DPP ladder - and only the wave sums meet. ----
        const bool from_gran = S.in_kind == 1;
        uint32_t tag = 0;
        const chain_gu64* g = nullptr;
        if (from_gran) {
          if (!need_gen(s)) return;
          tag = gen * 16u + (uint32_t)S.src + 1u;
          g = (const chain_gu64*)(args.gran + args.st[S.src].gran_off);
        }
        const bool lab_nosweep = (args.lab & 4) != 0;
        const int nslots = nc * 4;
        const int nwv = norm ? S.norm_nwv : 1, nai = norm ? S.norm_nai : 1;
        // the rows a later stage of this CU adds as its residual (the output of stage S.src): kept as they pass
        int stash_n0 = 0, stash_nr = 0, stash_off = 0;
        if (from_gran && S.stash_for >= 0) {
          const ChainStage& S2 = args.st[S.stash_for];
          int u0, u1;
          cw.task_range(S2.tasks, u0, u1);
          stash_n0 = 2 * u0;
          stash_nr = 2 * (u1 - u0);
          if (stash_n0 + stash_nr > S2.N) stash_nr = S2.N - stash_n0;
          stash_off = S2.stash_off;
        }
        u32x4* tile = reinterpret_cast<u32x4*>(smem + a_off);
        bool swept_any = false;
        if (from_gran) {
          // granules -> the tile, pass by pass (1024 granules: 16 coalesced relaxed agent-scope 8-byte loads per lane, pass p by
          // consumer p % NCONS; kept when every tag matches, read again after a nap otherwise): one ds_write_b32 per granule,
          // and without a norm that IS the staged input
          const int ng = lab_nosweep ? 0 : S.K / 2;
          const int npass = (S.K / 2 + 1023) / 1024;
          for (int p = ci; p < npass; p += NCONS) {
            if (args.thin && !swept_any && lane == 0)
              __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(smem) + CL_SWEEPING, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            swept_any = true;
            unsigned n_ = 0;
            unsigned long long t_ = 0;
            const unsigned long long tg0 = now();
            for (;;) {
              unsigned long long x[16];
#pragma unroll
              for (int k = 0; k < 16; ++k) {
                const int gi = p * 1024 + k * 64 + lane;
                x[k] = gi < ng ? __hip_atomic_load(g + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
              }
              bool ok = true;
#pragma unroll
              for (int k = 0; k < 16; ++k) ok &= (uint32_t)(x[k] >> 32) == tag;
              if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) reinterpret_cast<uint32_t*>(smem + a_off + p * 4096)[k * 64 + lane] = (uint32_t)x[k];
                break;
              }
              for (int i = 0; i < args.sweep_sleep; ++i) __builtin_amdgcn_s_sleep(8);       // ~0.2 us each
              if (cw.expired(n_, t_)) {
                cw.fail(CE_SWEEP, s);
                return;
              }
            }
            acc_t[4] += now() - tg0;
            for (int i = lane; i < stash_nr; i += 64) {
              const int n = stash_n0 + i - p * 2048;
              if (n >= 0 && n < 2048) reinterpret_cast<uint16_t*>(smem + stash_off)[i] = reinterpret_cast<const uint16_t*>(smem + a_off + p * 4096)[n];
            }
          }
          if (args.thin && swept_any && lane == 0)
            __hip_atomic_fetch_sub(reinterpret_cast<uint32_t*>(smem) + CL_SWEEPING, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // lane chunks past the last granule (K not a multiple of 2048): zeros
          for (int i = (S.K / 8) + ci * 64 + lane; i < nslots * 64; i += NCONS * 64) tile[i] = u32x4{0u, 0u, 0u, 0u};
          mark(1);
          if (norm && !sync(1)) return;             // the whole row is in the tile
          mark(2);
        }
        const int nunits = (from_gran && !norm) ? 0 : norm ? nwv : nslots;      // norm: a unit = a virtual wave's slots (at most 3); else one slot
        for (int unit = ci; unit < nunits; unit += NCONS) {
          constexpr int NS = 3;
          u32x4 item[NS];
          int slot_of[NS];
          bool have[NS];
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            slot_of[j] = norm ? unit + j * nwv : unit;
            have[j] = j < (norm ? nai : 1) && slot_of[j] < nslots;
            item[j] = u32x4{0u, 0u, 0u, 0u};
          }
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            if (!have[j]) continue;
            const int cl = (slot_of[j] >> 2) * 64 + lane;
            const long it = (long)cl * 4 + (slot_of[j] & 3);               // item index: 8 elements
            if (cl >= cpr) continue;                                        // past K: zeros
            item[j] = from_gran ? tile[it] : CHAIN_G(u32x4, S.A)[it];
          }
          if (!norm) {
            const int cl = (slot_of[0] >> 2) * 64 + lane;
            tile[(long)cl * 4 + (slot_of[0] & 3)] = item[0];               // (zeros past K)
            continue;
          }
          // the launch's thread (wave `unit`, lane): sum of squares over its items in item order, the wave's DPP ladder
          float ssq = 0.f;
          u32x4 nwv_item[NS];
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            float part = 0.f;
            nwv_item[j] = u32x4{0u, 0u, 0u, 0u};
            const int cl = (slot_of[j] >> 2) * 64 + lane;
            if (have[j] && cl < cpr) {
#pragma unroll
              for (int e = 0; e < 4; ++e) part = __builtin_amdgcn_fdot2(as_h2(item[j][e]), as_h2(item[j][e]), part, false);
              nwv_item[j] = CHAIN_G(u32x4, S.norm_weight)[(long)cl * 4 + (slot_of[j] & 3)];
            }
            if (j < nai) ssq += part;
          }
          const float ws = wave_sum_l63(ssq);
          if (lane == 63) reinterpret_cast<float*>(smem)[CL_WSUM + unit] = ws;
          if (unit + NCONS < nunits) {
            // (more than NCONS virtual waves: this consumer has a second one - park this one's items in the tile, unscaled,
            // and scale them in place behind the meeting)
#pragma unroll
            for (int j = 0; j < NS; ++j)
              if (have[j]) tile[(long)((slot_of[j] >> 2) * 64 + lane) * 4 + (slot_of[j] & 3)] = item[j];
            continue;
          }
          mark(3);
          if (!sync(0)) return;                     // every wave's sum of squares is in LDS
          mark(4);
          float tot = reinterpret_cast<const float*>(smem)[CL_WSUM];
          for (int w = 1; w < nwv; ++w) tot += reinterpret_cast<const float*>(smem)[CL_WSUM + w];
          const float r = rsqrtf(tot * S.norm_inv_k + S.norm_eps);
          for (int un2 = ci; un2 <= unit; un2 += NCONS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
              const int sl2 = un2 + j * nwv;
              if (!(j < nai && sl2 < nslots)) continue;
              const int cl = (sl2 >> 2) * 64 + lane;
              const long it = (long)cl * 4 + (sl2 & 3);
              u32x4 x = item[j], wgt = nwv_item[j];
              if (un2 != unit) {                     // a parked unit: its items and weights again
                x = tile[it];
                wgt = cl < cpr ? CHAIN_G(u32x4, S.norm_weight)[it] : u32x4{0u, 0u, 0u, 0u};
              }
              u32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const half2_t xx = as_h2(x[e]);
                const half2_t h = {(half_t)((float)xx[0] * r), (half_t)((float)xx[1] * r)};
                o[e] = cl < cpr ? as_u32(as_h2(wgt[e]) * h) : 0u;
              }
              tile[it] = o;
            }
          }
        }
        if (norm && ci >= nunits) {
          if (!sync(0)) return;                     // (a consumer without a unit still attends the meeting)
        }
      } else if (S.in_kind == 0) {
        // the caller's vector: plain loads of this consumer's chunks into their regions (natural order)
        for (int c = ci; c < nc; c += NCONS) {
          u32x4 x[4][IVW];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = c * 256 + q * 64 + lane;
            const bool valid = (i >> 2) < cpr;
#pragma unroll
            for (int v = 0; v < IVW; ++v) x[q][v] = CHAIN_G(u32x4, S.A)[(long)(valid ? i : 0) * IVW + v];
          }
          unsigned char* dstp = smem + a_off + (long)c * (64 * E * 2);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int v = 0; v < IVW; ++v) reinterpret_cast<u32x4*>(dstp + (long)(q * 64 + lane) * (EPW * 2))[v] = x[q][v];
          if (norm) chain_norm_parts<P>(smem, args, S, lane, c);
          else chain_stage_chunk<P, false>(smem, a_off, sa_off, cpr, c, lane, 0.f, nullptr);
        }
      } else {
        // granules of stage S.src: relaxed agent-scope 8-byte loads, 16 per lane and pass; a pass is kept when every one of its
        // tags matches, an incomplete one is read again after a nap; a chunk is staged when its passes are in
        if (!need_gen(s)) return;
        const uint32_t tag = gen * 16u + (uint32_t)S.src + 1u;
        const int ng = (args.lab & 4) ? 0 : S.K / 2;
        const chain_gu64* g = (const chain_gu64*)(args.gran + args.st[S.src].gran_off);
        const bool mine = ci < nc;
        if (mine && args.thin && lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(smem) + CL_SWEEPING, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // the rows a later stage of this CU adds as its residual (the output of stage S.src): kept as they pass
        int stash_n0 = 0, stash_nr = 0, stash_off = 0;
        if (mine && S.stash_for >= 0) {
          const ChainStage& S2 = args.st[S.stash_for];
          int u0, u1;
          cw.task_range(S2.tasks, u0, u1);
          stash_n0 = 2 * u0;
          stash_nr = 2 * (u1 - u0);
          if (stash_n0 + stash_nr > S2.N) stash_nr = S2.N - stash_n0;
          stash_off = S2.stash_off;
        }
        for (int c = ci; c < nc; c += NCONS) {
          unsigned char* dstp = smem + a_off + (long)c * (64 * E * 2);
          for (int h = 0; h < PPC; ++h) {
            const int p = c * PPC + h;
            unsigned n_ = 0;
            unsigned long long t_ = 0;
            const unsigned long long tg0 = now();
            for (;;) {
              unsigned long long x[16];
#pragma unroll
              for (int k = 0; k < 16; ++k) {
                const int gi = p * 1024 + k * 64 + lane;
                x[k] = gi < ng ? __hip_atomic_load(g + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
              }
              bool ok = true;
#pragma unroll
              for (int k = 0; k < 16; ++k) ok &= (uint32_t)(x[k] >> 32) == tag;
              if (__all(ok)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) reinterpret_cast<uint32_t*>(dstp + h * 4096)[k * 64 + lane] = (uint32_t)x[k];
                break;
              }
              for (int i = 0; i < args.sweep_sleep; ++i) __builtin_amdgcn_s_sleep(8);       // ~0.2 us each
              if (cw.expired(n_, t_)) {
                cw.fail(CE_SWEEP, s);
                return;
              }
            }
            acc_t[4] += now() - tg0;
            for (int i = lane; i < stash_nr; i += 64) {
              const int n = stash_n0 + i - p * 2048;
              if (n >= 0 && n < 2048) reinterpret_cast<uint16_t*>(smem + stash_off)[i] = reinterpret_cast<const uint16_t*>(dstp + h * 4096)[n];
            }
          }
          if (norm) chain_norm_parts<P>(smem, args, S, lane, c);
          else chain_stage_chunk<P, false>(smem, a_off, sa_off, cpr, c, lane, 0.f, nullptr);
        }
        if (mine && args.thin && lane == 0) __hip_atomic_fetch_sub(reinterpret_cast<uint32_t*>(smem) + CL_SWEEPING, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (!kChainNatural<P> && norm) {
        // the norm's weight for this consumer's chunk (host: at most NCONS lane chunks under a norm - one per consumer): asked
        // for now, used behind the meeting and the wave ladders
        u32x4 nwr[4][IVW];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = ci * 256 + q * 64 + lane;
          const bool valid = ci < nc && (i >> 2) < cpr;
#pragma unroll
          for (int v = 0; v < IVW; ++v) nwr[q][v] = CHAIN_G(u32x4, S.norm_weight)[(long)(valid ? i : 0) * IVW + v];
        }
        if (!sync(0)) return;                       // the whole row and its per-item sums of squares are in LDS
        if (ci < nc) {
          const float r = chain_norm_rinv(smem, args, S, lane);
          mark(3);
          chain_stage_chunk<P, true>(smem, a_off, sa_off, cpr, ci, lane, r, nwr);
          mark(4);
        }
      }
      mark(5);
      if (!sync(3)) return;                         // the tile is complete
      mark(6);
      // every workgroup of this launch has read the generation by now (its granules are here): the next launch's
      if (S.in_kind == 1 && s == args.bump_stage && cw.b == 0 && ci == 0 && lane == 0)
        __hip_atomic_store((chain_gu32*)args.ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cw.stamp(5 + 3 * s);
      acc_t[2] += now() - tst0;
    }
    // ---- this consumer's tasks of the stage: task k of the CU's range belongs to lane k % NL, the lane's j-th task to its
    // consumer j % CPL ----
    ChainTaskCtx X;
    X.nc = S.nc; X.cpr = S.cpr; X.kg = S.kg; X.gq_shift = S.gq_shift; X.N = S.N; X.n0 = n0;
    X.gq_magic = S.gq_magic; X.flip = S.flip;
    X.a_off = S.a_off; X.sa_off = S.sa_off; X.ring_off = args.ring_off + ln * RING * 1024; X.ring_units = RING;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      X.sc_rel[op] = 0;
      X.z_rel[op] = 0;
      if (MODE != MD_NONE && (op == 0 || S.pair)) {
        const unsigned long long sb = (unsigned long long)(S.scale[op]) + (unsigned long long)((long)n0 * S.kg * 2);
        X.sc_rel[op] = S.sc_off + (op * NTENS) * S.sc_units * 1024 + (int)(sb & 15ull);
        if (NTENS == 2) {
          const unsigned long long zb = (unsigned long long)(S.zeros[op]) + (unsigned long long)((long)n0 * S.kg * 2);
          X.z_rel[op] = S.sc_off + (op * NTENS + 1) * S.sc_units * 1024 + (int)(zb & 15ull);
        }
      }
    }
    X.zint = (float)S.zint;
    X.has_bias = S.has_bias;
    X.has_res = (S.residual != nullptr || S.res_stage >= 0) ? 1 : 0;
    X.stash_off = S.stash_off;
    X.bias[0] = S.bias[0]; X.bias[1] = S.bias[1];
    X.residual = S.residual;
    X.C = S.C;
    X.gran = nullptr;
    X.tag = 0;
    if (S.publish) {
      if (!need_gen(s)) return;
      X.tag = gen * 16u + (uint32_t)s + 1u;
      X.gran = args.gran + S.gran_off;
    }
    cyc(s, 7);
    const int un = S.un, pair = S.pair;
    // the stage's scale / zeros blocks ride at the head of lane 0's stream
    if (S.nsc > 0 && !(args.lab & 2) && !cw.wait_ge(CL_LANDED0, (uint32_t)(useq0_base + S.nsc), CE_WAIT_LANDED, s)) return;
    const int need_base = useq_base + (ln == 0 ? S.nsc : 0);
    const int ntl = nt > ln ? (nt - ln + NL - 1) / NL : 0;      // tasks of this stage that fall to this lane: ln, ln + NL, ...
    int jstep = CPL * un;
    while (jstep >= RING) jstep -= RING;
    int rpos = (rseq_base + sub * un) % RING;
    for (int j = sub; j < ntl; j += CPL) {
      const int k = ln + j * NL;
      const unsigned long long tw1 = now();
      const int need0 = need_base + j * un, rseq0 = rseq_base + j * un;
      if (args.lab & 1) {
        if (!(args.lab & 2) && !cw.wait_ge(CL_LANDED0 + ln, (uint32_t)(need0 + un), CE_WAIT_LANDED, s)) return;
        if (X.gran && lane == 63) __hip_atomic_store((chain_gu64*)(X.gran + t0 + k), (unsigned long long)X.tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (pair) {
        if (!chain_task<P, 4>(cw, X, t0 + k, rpos, CL_LANDED0 + ln, need0, CL_NEXT0 + 4 * ln + sub, rseq0, s, (args.lab & 2) != 0)) return;
      } else {
        if (!chain_task<P, 2>(cw, X, t0 + k, rpos, CL_LANDED0 + ln, need0, CL_NEXT0 + 4 * ln + sub, rseq0, s, (args.lab & 2) != 0)) return;
      }
      // this consumer's next unfinished task of the lane (its ring slots went back chunk by chunk)
      int next = rseq_base + (j + CPL) * un;
      if (j + CPL >= ntl) next = rseq_base + ntl * un;
      chain_lds_st(smem, CL_NEXT0 + 4 * ln + sub, (uint32_t)next);
      acc_t[1] += now() - tw1;
      rpos += jstep;
      if (rpos >= RING) rpos -= RING;
    }
    // (also when no task of this stage fell to this consumer: its frontier still moves past the stage)
    chain_lds_st(smem, CL_NEXT0 + 4 * ln + sub, (uint32_t)(rseq_base + ntl * un));
    cyc(s, 8);
    cw.stamp(6 + 3 * s);
    rseq_base += ntl * un;
    useq_base += (ln == 0 ? S.nsc : 0) + ntl * un;
    useq0_base += S.nsc + (nt > 0 ? (nt + NL - 1) / NL : 0) * un;
  }
  chain_lds_st(smem, CL_NEXT0 + 4 * ln + sub, 0x7fffffffu);
  chain_lds_st(smem, CL_CSTAGE0 + ci, 0x7fffffffu);
  if (timing && lane == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) args.trace[((long)cw.b * 16 + cw.wave) * 64 + 26 + i] = acc_t[i];
  }
}

// nlanes loaders (waves 0 .. NL-1: one per SIMD) + nlanes * cpl consumers: at most 16 waves, 128 registers each
template <int BITS, int LAYOUT, int MODE>
__global__ void __launch_bounds__(1024) wq_chain_kernel(const ChainArgs args) {
  using P = GemvxPolicy<BITS, LAYOUT, MODE, 1, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  ChainWave cw;
  cw.smem = smem_raw;
  cw.args = &args;
  cw.lane = tid & 63;
  cw.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cw.b = (int)blockIdx.x;
  cw.G = (int)gridDim.x;
  cw.timeout = args.timeout_ticks;
  if (tid < CL_WORDS) {
    uint32_t v = 0u;
    const int ncons = args.nlanes * args.cpl;
    if (tid >= CL_NEXT0 && tid < CL_NEXT0 + 16 && ((tid - CL_NEXT0) >> 2 >= args.nlanes || ((tid - CL_NEXT0) & 3) >= args.cpl)) v = 0x7fffffffu;
    if (tid >= CL_CSTAGE0 && tid < CL_CSTAGE0 + 16 && tid - CL_CSTAGE0 >= ncons) v = 0x7fffffffu;
    reinterpret_cast<uint32_t*>(smem_raw)[tid] = v;
  }
  __syncthreads();
  cw.stamp(0);
  if (cw.wave < args.nlanes) chain_loader<P>(cw);
  else {
    if (args.lab & 32) __builtin_amdgcn_s_setprio(3);      // lab: the consumers above the loaders in issue priority
    chain_consumer<P>(cw);
  }
  cw.stamp(3);
}

typedef void (*chain_fn)(const ChainArgs);
chain_fn pick_chain(int bits, int layout, int mode);

}  // namespace wqaa
