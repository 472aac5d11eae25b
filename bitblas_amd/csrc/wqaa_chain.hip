// wqaa_chain.hip - host side of wqaa_matmul_chain: validation, the plan of the persistent launch (stage descriptors, LDS
// layout, hand-off scratch), launch - and the launch-by-launch form of the same chain for everything the fused member does
// not cover.  The kernel: wqaa_chain_kernel.h.
#include "wqaa_chain_kernel.h"

#include <mutex>
#include <vector>

namespace wqaa {

// member table: one kernel per (bits, layout, scale / zeros mode) - every stage of a chain shares the format
chain_fn pick_chain(int bits, int layout, int mode) {
#define WQAA_CH(B, L)                                                   \
  switch (mode) {                                                       \
    case MD_NONE: return wq_chain_kernel<B, L, MD_NONE>;                \
    case MD_S: return wq_chain_kernel<B, L, MD_S>;                      \
    case MD_ZO: return wq_chain_kernel<B, L, MD_ZO>;                    \
    case MD_ZR: return wq_chain_kernel<B, L, MD_ZR>;                    \
  }                                                                     \
  return nullptr;
  if (bits == 4 && layout == LAYOUT_LOP3) { WQAA_CH(4, LAYOUT_LOP3) }
  if (bits == 4 && layout == LAYOUT_PLAIN) { WQAA_CH(4, LAYOUT_PLAIN) }
  if (bits == 2 && layout == LAYOUT_LOP3) { WQAA_CH(2, LAYOUT_LOP3) }
  if (bits == 2 && layout == LAYOUT_PLAIN) { WQAA_CH(2, LAYOUT_PLAIN) }
#undef WQAA_CH
  return nullptr;
}

static int desc_mode(const wqaa_matmul_desc& d) {
  return !d.with_scaling ? MD_NONE
         : d.zeros_mode == WQAA_Z_ORIGINAL ? MD_ZO
         : d.zeros_mode == WQAA_Z_RESCALE  ? MD_ZR
         : d.zeros_mode == WQAA_Z_QUANTIZED ? MD_ZQ
                                            : MD_S;
}

// ---- hand-off scratch: one slab per (device, stream): [0, 256) control words (generation, first error), granules, the
// outputs of items the caller gave no C (launch-by-launch form), the lab's time stamps.  Zeroed when allocated (tags of a
// generation never equal 0-initialised granules: tag = generation * 16 + stage + 1); grown = replaced + zeroed, outside
// stream capture only. ----
struct ChainSlab {
  int dev;
  hipStream_t stream;
  unsigned char* ptr;
  size_t bytes;
  size_t trace_words;   // of the last traced launch
  size_t trace_off;
};
static std::vector<ChainSlab> g_chain_ws;
static std::vector<void*> g_chain_retired;
static std::mutex g_chain_mu;
constexpr size_t kChainCtlBytes = 256;

// streams first seen DURING capture (torch captures on a side stream of its own) take a spare slab: allocated and cleared
// next to the device's first slab, outside capture
constexpr int kChainSpares = 8;
constexpr size_t kChainSlabBytes = 1u << 20;
struct ChainSpare {
  int dev;
  unsigned char* ptr;
};
static std::vector<ChainSpare> g_chain_spares;

static ChainSlab* chain_slab(hipStream_t stream, size_t bytes, bool create) {
  const int dev = current_device();
  if (dev < 0) return nullptr;
  ChainSlab* slab = nullptr;
  for (auto& w : g_chain_ws)
    if (w.dev == dev && w.stream == stream) { slab = &w; break; }
  if (!create) return slab;
  if (slab && slab->bytes >= bytes) return slab;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) {
    if (!slab && bytes <= kChainSlabBytes) {
      for (size_t i = 0; i < g_chain_spares.size(); ++i)
        if (g_chain_spares[i].dev == dev) {
          g_chain_ws.push_back(ChainSlab{dev, stream, g_chain_spares[i].ptr, kChainSlabBytes, 0, 0});
          g_chain_spares.erase(g_chain_spares.begin() + (long)i);
          return &g_chain_ws.back();
        }
    }
    set_error(WQAA_ERR_LAUNCH, "matmul_chain: the hand-off scratch of this stream has to be allocated (%zu B), which cannot happen during "
              "stream capture: run a chain once outside capture first (that also sets %d spare slabs aside for capturing streams)", bytes, kChainSpares);
    return nullptr;
  }
  size_t want = bytes < kChainSlabBytes ? kChainSlabBytes : bytes;
  if (slab && want < 2 * slab->bytes) want = 2 * slab->bytes;
  bool have_spares = false;
  for (auto& sp : g_chain_spares) have_spares = have_spares || sp.dev == dev;
  bool first_of_dev = !have_spares;
  for (auto& w : g_chain_ws) first_of_dev = first_of_dev && w.dev != dev;
  const int nalloc = 1 + (first_of_dev ? kChainSpares : 0);
  void* got[1 + kChainSpares] = {};
  for (int i = 0; i < nalloc; ++i) {
    const size_t sz = i == 0 ? want : kChainSlabBytes;
    // stream-ordered clear, then one wait: a spare is used by ANOTHER stream later
    if (hipMalloc(&got[i], sz) != hipSuccess || hipMemsetAsync(got[i], 0, sz, stream) != hipSuccess) {
      (void)hipGetLastError();
      for (int j = 0; j <= i; ++j)
        if (got[j]) (void)hipFree(got[j]);
      set_error(WQAA_ERR_LAUNCH, "matmul_chain: cannot allocate %zu B of hand-off scratch", sz);
      return nullptr;
    }
  }
  if (nalloc > 1 && hipStreamSynchronize(stream) != hipSuccess) (void)hipGetLastError();
  for (int i = 1; i < nalloc; ++i) g_chain_spares.push_back(ChainSpare{dev, reinterpret_cast<unsigned char*>(got[i])});
  if (slab) {
    g_chain_retired.push_back(slab->ptr);
    slab->ptr = reinterpret_cast<unsigned char*>(got[0]);
    slab->bytes = want;
    slab->trace_words = 0;
  } else {
    g_chain_ws.push_back(ChainSlab{dev, stream, reinterpret_cast<unsigned char*>(got[0]), want, 0, 0});
    slab = &g_chain_ws.back();
  }
  return slab;
}

// switches (A/B and lab aids; all plan-time like every WQAA_* variable: read again when wqaa_select / wqaa_chain_plan bump the epoch)
struct ChainKnobs {
  int fuse, lanes, cpl, ring, thin, sweep_sleep, trace, lab;
  unsigned timeout_ticks;
};
static const ChainKnobs& chain_knobs() {
  static thread_local ChainKnobs k;
  static thread_local unsigned seen = ~0u;
  const unsigned ep = g_plan_epoch.load(std::memory_order_relaxed);
  if (ep != seen) {
    auto geti = [](const char* name, int dflt) {
      const char* f = getenv(name);
      return f ? atoi(f) : dflt;
    };
    // Round 5: the persistent member is OPT-IN (WQAA_CHAIN_FUSE=1).  `wqaa_matmul_chain` is DEFINED as the launches it stands for and
    // runs them by default: they are faster (26.0 vs 38.5 us per decoder-layer tail, profiles/r04_chain_lab.txt) and need nothing
    // from the rest of the chip, whereas the persistent launch spin-waits on granules other workgroups write - its grid of one
    // 160 KiB workgroup per CU is only co-resident on an otherwise idle device (two chains on two streams, a CU mask or
    // any kernel holding LDS leave part of it queued; every wait then ends in its 250 ms bound with an error code only
    // `wqaa_debug_chain_status` reads).  Whoever opts in owns that precondition.
    k.fuse = geti("WQAA_CHAIN_FUSE", 0) != 0;
    // (round 4's lab aids WQAA_CHAIN_LANES / _CPL / _RING / _THIN / _SWEEP_SLEEP / _LAB are gone with round 5's prune: the values
    // they settled on - profiles/r04_chain_lab.txt)
    k.lanes = 0;
    k.cpl = 0;
    k.ring = 0;
    k.thin = 1;
    k.sweep_sleep = 2;
    k.trace = geti("WQAA_CHAIN_TRACE", 0);
    k.lab = 0;
    const int ms = geti("WQAA_CHAIN_TIMEOUT_MS", 250);
    k.timeout_ticks = (unsigned)(ms > 0 ? ms : 1) * 100000u;          // s_memrealtime runs at 100 MHz
    seen = ep;
  }
  return k;
}

struct ChainBuild {
  ChainArgs args;
  int bits, layout, mode, lanes, cpl;
  int lds_bytes;
  int grid;
  size_t gran_count;        // granules
  long stream_bytes;        // weight bytes a launch streams (plan report)
};

static inline int align16(int x) { return (x + 15) & ~15; }

// checks that make the chain ill-formed whatever runs it
static int chain_validate(const wqaa_chain_item* items, int count, int m) {
  if (!items || count < 1 || count > WQAA_CHAIN_MAX || m < 0) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_chain: bad arguments (items=%p count=%d m=%d; at most %d items)", (const void*)items, count, m, WQAA_CHAIN_MAX);
    return WQAA_ERR_BAD_DESC;
  }
  for (int i = 0; i < count; ++i) {
    const wqaa_chain_item& it = items[i];
    if (!it.desc || it.desc->struct_size != (int32_t)sizeof(wqaa_matmul_desc) || it.desc->N <= 0 || it.desc->K <= 0) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d has a missing or malformed descriptor", i);
      return WQAA_ERR_BAD_DESC;
    }
    const wqaa_matmul_desc& d = *it.desc;
    if (it.kind != 0 && it.kind != 1) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d kind %d (0 matmul, 1 gate / up pair)", i, it.kind);
      return WQAA_ERR_BAD_DESC;
    }
    const bool pair = it.kind == 1;
    if (!it.B || (d.with_scaling && !it.Scale) || (d.zeros_mode != WQAA_Z_NONE && !it.Zeros) || (d.with_bias && !it.Bias) ||
        (pair && (!it.B2 || (d.with_scaling && !it.Scale2) || (d.zeros_mode != WQAA_Z_NONE && !it.Zeros2) || (d.with_bias && !it.Bias2)))) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d has a null operand its descriptor requires", i);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.input_from >= i || it.residual_from >= i || (it.input_from < 0 && !it.A)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d reads input %d / residual %d: an earlier item, or -1 with the pointer", i, it.input_from,
                it.residual_from);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.input_from >= 0 && items[it.input_from].desc->N != d.K) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d (K = %d) reads the output of item %d (N = %d)", i, d.K, it.input_from,
                items[it.input_from].desc->N);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.residual_from >= 0 && items[it.residual_from].desc->N != d.N) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d (N = %d) adds the output of item %d (N = %d)", i, d.N, it.residual_from,
                items[it.residual_from].desc->N);
      return WQAA_ERR_BAD_DESC;
    }
    if (pair && (it.residual || it.residual_from >= 0)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d: a gate / up pair takes no residual", i);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.residual && it.residual_from >= 0) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d names two residuals", i);
      return WQAA_ERR_BAD_DESC;
    }
    if ((it.residual || it.residual_from >= 0) && it.norm_weight) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d: RMSNorm input and residual add on one operator (wqaa_matmul_ex has no such launch)", i);
      return WQAA_ERR_UNSUPPORTED;
    }
  }
  return WQAA_OK;
}

// the fused member's plan, or WQAA_ERR_UNSUPPORTED (with the reason in the error string): the caller runs the launches
static int chain_build(const wqaa_chain_item* items, int count, int m, ChainBuild* out) {
  memset(out, 0, sizeof(*out));
  ChainArgs& A = out->args;
  if (m != 1) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: the persistent member takes m = 1 (got %d)", m);
    return WQAA_ERR_UNSUPPORTED;
  }
  const ChainKnobs& knobs = chain_knobs();
  if (!knobs.fuse) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: WQAA_CHAIN_FUSE=0");
    return WQAA_ERR_UNSUPPORTED;
  }
  const int cus = device_info().ok ? device_info().cus : 256;
  const wqaa_matmul_desc& d0 = *items[0].desc;
  out->bits = d0.w_bits;
  out->layout = d0.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN;
  out->mode = desc_mode(d0);
  out->grid = cus;
  if (!pick_chain(out->bits, out->layout, out->mode)) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: no persistent member for %d-bit weights, layout %d, scale / zeros mode %d", out->bits, out->layout, out->mode);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int E = 128 / out->bits;
  int gen_of[kChainMaxStages];               // input generation of a stage (which of the two LDS tiles)
  int first_of_gen[kChainMaxStages + 1];
  int ngen = 0;
  size_t gran = 0;
  for (int i = 0; i < count; ++i) {
    const wqaa_chain_item& it = items[i];
    const wqaa_matmul_desc& d = *it.desc;
    ChainStage& S = A.st[i];
    const bool pair = it.kind == 1;
    const bool has_res = it.residual != nullptr || it.residual_from >= 0;
    if (d.w_bits != out->bits || (d.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN) != out->layout || desc_mode(d) != out->mode ||
        d.w_format != d0.w_format) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d has another weight format than item 0 (one format per persistent launch)", i);
      return WQAA_ERR_UNSUPPORTED;
    }
    GemvxArgs ga;
    int kw = 1, nw = 8, nai = 2;
    const int pro = it.norm_weight ? (pair ? 4 : 3) : pair ? 2 : has_res ? 1 : 0;
    int st = gemvx_chain_geometry(d, m, pro, pair, &ga, &kw, &nw, &nai);
    if (st != WQAA_OK) return st;
    // a PLAIN item (no norm, no residual, not a pair) stands for `wqaa_matmul`, which takes the exact-product family only where
    // gemvx_eligible says so (not for strict_reference descriptors, not under WQAA_GEMVX=0, not behind its fences): anywhere else
    // the launches' bits are the rounding family's and the persistent member - exact products always - would differ
    if (pro == 0 && !gemvx_eligible(d, m)) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d's single launch is a per-element-rounding member (strict_reference, or outside the exact-product family's fences): other bits than the persistent member's", i);
      return WQAA_ERR_UNSUPPORTED;
    }
    if (kw != 1) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d's single launch splits K over %d waves (another summation order than the chain's)", i, kw);
      return WQAA_ERR_UNSUPPORTED;
    }
    if (d.N < 2 * cus || (long)((d.N + 1) / 2) * cus >= (1l << 32)) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d: N = %d outside [2 x CUs, 2^32 / CUs)", i, d.N);
      return WQAA_ERR_UNSUPPORTED;
    }
    S.A = it.A;
    S.B[0] = it.B; S.scale[0] = it.Scale; S.zeros[0] = it.Zeros; S.bias[0] = it.Bias;
    S.B[1] = pair ? it.B2 : nullptr; S.scale[1] = pair ? it.Scale2 : nullptr; S.zeros[1] = pair ? it.Zeros2 : nullptr; S.bias[1] = pair ? it.Bias2 : nullptr;
    S.residual = it.residual;
    S.norm_weight = it.norm_weight;
    S.C = it.C;
    S.norm_eps = it.norm_eps;
    S.norm_inv_k = 1.f / (float)d.K;
    S.N = d.N; S.K = d.K; S.kg = ga.kg; S.gq_shift = ga.gq_shift; S.gq_magic = ga.gq_magic;
    S.nc = ga.nc; S.cpr = ga.cpr;
    if (ga.row_bytes > 0x7fffffffl) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d row too long", i);
      return WQAA_ERR_UNSUPPORTED;
    }
    S.row_bytes = (int)ga.row_bytes;
    S.zint = ga.zint; S.flip = ga.flip; S.has_bias = d.with_bias;
    S.pair = pair ? 1 : 0;
    S.publish = 0;
    S.res_stage = it.residual_from;
    S.stash_for = -1;
    S.norm_nwv = nw; S.norm_nai = nai;
    S.tasks = (d.N + 1) / 2;
    S.un = (pair ? 4 : 2) * S.nc;
    // input kind: the previous item's staged tile serves when it is the same vector through the same norm
    S.src = it.input_from;
    S.in_kind = it.input_from >= 0 ? 1 : 0;
    if (i > 0) {
      const wqaa_chain_item& pv = items[i - 1];
      const ChainStage& PS = A.st[i - 1];
      const bool same_vec = it.input_from >= 0 ? pv.input_from == it.input_from : (pv.input_from < 0 && pv.A == it.A);
      const bool same_norm = pv.norm_weight == it.norm_weight && (!it.norm_weight || (pv.norm_eps == it.norm_eps && PS.norm_nwv == nw && PS.norm_nai == nai));
      if (same_vec && same_norm && pv.desc->K == d.K) S.in_kind = 2;
    }
    if (it.norm_weight) {
      // (the single launch's own limit - the row within the items its workgroup loads ahead - was checked by its selector)
      if (S.nc > kChainMaxLanes * kChainMaxCpl || nw > 16) {
        set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d: RMSNorm input of K = %d", i, d.K);
        return WQAA_ERR_UNSUPPORTED;
      }
    }
    if (S.in_kind != 2) {
      first_of_gen[ngen] = i;
      gen_of[i] = ngen++;
    } else {
      gen_of[i] = ngen - 1;
    }
    out->stream_bytes += (long)(pair ? 2 : 1) * d.N * ga.row_bytes;
  }
  // edges: who publishes, where the generation is bumped, which sweep keeps a later stage's residual rows
  A.bump_stage = -1;
  for (int i = 0; i < count; ++i) {
    ChainStage& S = A.st[i];
    if (S.in_kind == 1) {
      A.st[S.src].publish = 1;
      A.bump_stage = i;
    } else if (S.in_kind == 2 && S.src >= 0) {
      A.st[S.src].publish = 1;
    }
  }
  for (int i = 0; i < count; ++i) {
    ChainStage& S = A.st[i];
    if (S.publish) {
      S.gran_off = (int)gran;
      gran += (size_t)S.tasks;
    }
  }
  for (int i = 0; i < count; ++i) {
    ChainStage& S = A.st[i];
    if (S.res_stage < 0) continue;
    int sweeper = -1;
    for (int j = S.res_stage + 1; j <= i; ++j)
      if (A.st[j].in_kind == 1 && A.st[j].src == S.res_stage) { sweeper = j; break; }
    if (sweeper < 0 || A.st[sweeper].stash_for >= 0) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d adds the output of item %d, which no item up to it reads as its input (the residual "
                "rows are kept while that input is swept)", i, S.res_stage);
      return WQAA_ERR_UNSUPPORTED;
    }
    A.st[sweeper].stash_for = i;
    const int rows_max = 2 * ((S.tasks + cus - 1) / cus);
    if (rows_max > kChainStashMaxRows) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d: %d residual rows per CU", i, rows_max);
      return WQAA_ERR_UNSUPPORTED;
    }
  }
  // ---- LDS layout ----
  constexpr int NTENS_MAX = 2;
  const int ntens = (out->mode == MD_ZO || out->mode == MD_ZR) ? 2 : out->mode == MD_NONE ? 0 : 1;
  (void)NTENS_MAX;
  int off = CL_WORDS * 4;
  for (int i = 0; i < count; ++i)
    if (A.st[i].res_stage >= 0) {
      A.st[i].stash_off = off;
      off += kChainStashMaxRows * 2;
    }
  // the norm's per-item partial sums (its raw row sits in the tile's own region until it is staged in place)
  int parts_bytes = 0, norm_nc_max = 0;
  for (int i = 0; i < count; ++i) {
    const ChainStage& S = A.st[i];
    if (S.in_kind == 2 || !S.norm_weight) continue;
    if (S.nc > norm_nc_max) norm_nc_max = S.nc;
    if (S.nc * 4 * 256 > parts_bytes) parts_bytes = S.nc * 4 * 256;
  }
  A.parts_off = off;
  off += parts_bytes;
  // scale / zeros blocks: two areas by stage parity
  int sc_size[2] = {0, 0};
  for (int i = 0; i < count; ++i) {
    ChainStage& S = A.st[i];
    const int rows_max = 2 * ((S.tasks + cus - 1) / cus);
    S.sc_units = ntens ? (15 + rows_max * S.kg * 2 + 1023) / 1024 : 0;
    S.nsc = S.sc_units * ntens * (S.pair ? 2 : 1);
    if (S.nsc * 1024 > sc_size[i & 1]) sc_size[i & 1] = S.nsc * 1024;
  }
  int sc_base[2];
  sc_base[0] = off; off += sc_size[0];
  sc_base[1] = off; off += sc_size[1];
  for (int i = 0; i < count; ++i) A.st[i].sc_off = sc_base[i & 1];
  // staged input tiles: two buffers by input generation parity
  int act_size[2] = {0, 0};
  for (int i = 0; i < count; ++i) {
    const ChainStage& S = A.st[i];
    const int sz = align16(S.nc * 64 * E * 2) + align16(S.nc * 64 * 4);
    if (sz > act_size[gen_of[i] & 1]) act_size[gen_of[i] & 1] = sz;
  }
  int act_base[2];
  act_base[0] = off; off += act_size[0];
  act_base[1] = off; off += act_size[1];
  for (int i = 0; i < count; ++i) {
    ChainStage& S = A.st[i];
    S.a_off = act_base[gen_of[i] & 1];
    S.sa_off = S.a_off + align16(S.nc * 64 * E * 2);
    const int g = gen_of[i];
    S.wait_stage = (S.in_kind != 2 && g >= 2) ? first_of_gen[g - 1] : 0;
  }
  off = (off + 1023) & ~1023;
  A.ring_off = off;
  const int lds_total = 160 * 1024;
  int un_max = 0;
  for (int i = 0; i < count; ++i) un_max = A.st[i].un > un_max ? A.st[i].un : un_max;
  // lanes (loader + consumer + ring slice each): four feed the memory's rate (one loader wave issues ~8.4 GB/s of LDS-DMA); fewer
  // when a task's rows would not fit a quarter of the ring beside a fill in flight each way
  const int total_units = (lds_total - off) / 1024;
  int lanes = kChainMaxLanes, ring_units = 0;
  // two consumers per lane: same-call A/B on the Llama-2-7B decoder tail (tools/chain_lab.py, profiles/r04_chain_lab.txt): 1 / 2 / 3
  // consumers per lane 48.7 / 37.2 / 40.9 us
  int cpl = 2;
  if (knobs.cpl >= 1 && knobs.cpl <= kChainMaxCpl) cpl = knobs.cpl;                   // lab aid
  if (knobs.lanes >= 1 && knobs.lanes <= kChainMaxLanes) lanes = knobs.lanes;         // lab aid
  for (; lanes >= 1; lanes >>= 1) {
    ring_units = total_units / lanes;
    ring_units -= ring_units % kChainFill;
    if (knobs.ring >= 2 * kChainFill && knobs.ring < ring_units) ring_units = knobs.ring - knobs.ring % kChainFill;   // lab aid: a smaller ring
    if (ring_units >= un_max + 2 * kChainFill && ring_units >= 4 * kChainFill && norm_nc_max <= lanes * cpl) break;
    if (knobs.lanes >= 1) { lanes = 0; break; }
  }
  if (lanes < 1) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: %d KiB of LDS left for the weight rings (a task needs %d), %d lane chunks under a norm", total_units,
              un_max, norm_nc_max);
    return WQAA_ERR_UNSUPPORTED;
  }
  out->lanes = lanes;
  out->cpl = cpl;
  A.nlanes = lanes;
  A.cpl = cpl;
  A.g_shift = (out->grid & (out->grid - 1)) == 0 ? ilog2_exact(out->grid) : -1;
  A.ring_units = ring_units;
  out->lds_bytes = A.ring_off + lanes * ring_units * 1024;
  A.nstages = count;
  A.thin = knobs.thin;
  A.sweep_sleep = knobs.sweep_sleep;
  A.timeout_ticks = knobs.timeout_ticks;
  A.lab = knobs.lab;
  out->gran_count = gran;
  return WQAA_OK;
}

static void chain_plan_fill(const ChainBuild& b, const wqaa_chain_item* items, int count, wqaa_plan* plan) {
  if (!plan) return;
  memset(plan, 0, sizeof(*plan));
  plan->kernel_family = 1;
  plan->block_m = 1;
  plan->block_n = 2;
  plan->block_k = 64 * (128 / b.bits);
  plan->threads = 64 * b.lanes * (1 + b.cpl);
  plan->grid = b.grid;
  plan->rows_per_wave = 2;
  plan->batch_tile = 1;
  plan->pipeline_depth = b.args.ring_units;
  plan->split_k = 1;
  plan->lds_bytes = b.lds_bytes;
  char wd[24];
  short_wdtype(*items[0].desc, wd, sizeof(wd));
  int n = snprintf(plan->name, sizeof(plan->name), "chain_m1_%sx%s", short_dtype(items[0].desc->a_dtype), wd);
  for (int i = 0; i < count && n > 0 && n < (int)sizeof(plan->name) - 1; ++i) {
    const ChainStage& S = b.args.st[i];
    n += snprintf(plan->name + n, sizeof(plan->name) - n, "_%s%s%dx%d%s", S.norm_weight ? "n" : "", S.pair ? "p" : "", S.N, S.K,
                  (S.residual || S.res_stage >= 0) ? "r" : "");
  }
  if (n > 0 && n < (int)sizeof(plan->name) - 1) snprintf(plan->name + n, sizeof(plan->name) - n, "_l%dc%dring%d", b.lanes, b.cpl, b.args.ring_units);
}

// ---- launch by launch: the definition of the chain, and its form wherever the persistent member does not cover it ----
static int chain_by_launches(const wqaa_chain_item* items, int count, int m, hipStream_t stream) {
  // outputs nobody gave a buffer for live in the stream's scratch
  size_t tmp_bytes = 0;
  size_t tmp_off[WQAA_CHAIN_MAX];
  for (int i = 0; i < count; ++i) {
    tmp_off[i] = tmp_bytes;
    if (!items[i].C) tmp_bytes += ((size_t)m * items[i].desc->N * 2 + 255) & ~(size_t)255;
  }
  unsigned char* tmp = nullptr;
  if (tmp_bytes > (1u << 18)) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: %zu B of outputs without a buffer (pass C)", tmp_bytes);
    return WQAA_ERR_UNSUPPORTED;
  }
  if (tmp_bytes) {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    ChainSlab* slab = chain_slab(stream, kChainCtlBytes + (1u << 19) + (1u << 18), true);
    if (!slab) return WQAA_ERR_LAUNCH;
    tmp = slab->ptr + kChainCtlBytes + (1u << 19);
  }
  auto out_of = [&](int i) -> void* { return items[i].C ? items[i].C : (void*)(tmp + tmp_off[i]); };
  for (int i = 0; i < count; ++i) {
    const wqaa_chain_item& it = items[i];
    const void* in = it.input_from >= 0 ? out_of(it.input_from) : it.A;
    const void* res = it.residual_from >= 0 ? out_of(it.residual_from) : it.residual;
    wqaa_epilogue epi;
    memset(&epi, 0, sizeof(epi));
    epi.struct_size = (int32_t)sizeof(epi);
    int st;
    if (it.kind == 1) {
      wqaa_group_item g, u;
      memset(&g, 0, sizeof(g));
      memset(&u, 0, sizeof(u));
      g.desc = u.desc = it.desc;
      g.A = u.A = in;
      g.B = it.B; g.Scale = it.Scale; g.Zeros = it.Zeros; g.Bias = it.Bias;
      u.B = it.B2; u.Scale = it.Scale2; u.Zeros = it.Zeros2; u.Bias = it.Bias2;
      if (it.norm_weight) {
        epi.flags = WQAA_EPI_RMSNORM_INPUT;
        epi.norm_weight = it.norm_weight;
        epi.norm_eps = it.norm_eps;
      }
      st = wqaa_matmul_gate_up(&g, &u, out_of(i), m, stream, it.norm_weight ? &epi : nullptr);
    } else if (it.norm_weight || res) {
      if (it.norm_weight) {
        epi.flags = WQAA_EPI_RMSNORM_INPUT;
        epi.norm_weight = it.norm_weight;
        epi.norm_eps = it.norm_eps;
      } else {
        epi.flags = WQAA_EPI_ADD_RESIDUAL;
        epi.residual = res;
      }
      st = wqaa_matmul_ex(it.desc, in, it.B, nullptr, it.Scale, it.Zeros, it.Bias, out_of(i), m, stream, &epi);
    } else {
      st = wqaa_matmul(it.desc, in, it.B, nullptr, it.Scale, it.Zeros, it.Bias, out_of(i), m, stream);
    }
    if (st != WQAA_OK) return st;
  }
  return WQAA_OK;
}

int chain_plan(const wqaa_chain_item* items, int count, int m, int* launches, wqaa_plan* plan) {
  int st = chain_validate(items, count, m);
  if (st != WQAA_OK) return st;
  if (plan) memset(plan, 0, sizeof(*plan));
  g_plan_epoch.fetch_add(1, std::memory_order_relaxed);
  ChainBuild b;
  if (m > 0 && chain_build(items, count, m, &b) == WQAA_OK) {
    if (launches) *launches = 1;
    chain_plan_fill(b, items, count, plan);
  } else {
    if (launches) *launches = count;
  }
  return WQAA_OK;
}

int chain_launch(const wqaa_chain_item* items, int count, int m, hipStream_t stream) {
  int st = chain_validate(items, count, m);
  if (st != WQAA_OK) return st;
  if (m == 0) return WQAA_OK;
  ChainBuild b;
  // (the plan is rebuilt per call: a few hundred nanoseconds per stage next to a 15-20 us launch; pointers differ per call)
  if (chain_build(items, count, m, &b) != WQAA_OK) return chain_by_launches(items, count, m, stream);
  const bool trace = chain_knobs().trace != 0;
  const size_t gran_bytes = (b.gran_count * 8 + 255) & ~(size_t)255;
  const size_t trace_words = trace ? (size_t)b.grid * 16 * 64 : 0;
  const size_t need = kChainCtlBytes + (1u << 19) + (1u << 18) + trace_words * 8;   // (a traced launch: 2 MiB more)
  if (gran_bytes > (1u << 19)) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: %zu B of granules", gran_bytes);
    return chain_by_launches(items, count, m, stream);
  }
  unsigned char* base;
  {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    ChainSlab* slab = chain_slab(stream, need, true);
    if (!slab) return WQAA_ERR_LAUNCH;
    base = slab->ptr;
    slab->trace_words = trace_words;
    slab->trace_off = kChainCtlBytes + (1u << 19) + (1u << 18);
  }
  b.args.ctl = reinterpret_cast<uint32_t*>(base);
  b.args.gran = reinterpret_cast<unsigned long long*>(base + kChainCtlBytes);
  b.args.trace = trace ? reinterpret_cast<unsigned long long*>(base + kChainCtlBytes + (1u << 19) + (1u << 18)) : nullptr;
  chain_fn fn = pick_chain(b.bits, b.layout, b.mode);
  void* params[] = {&b.args};
  dim3 grid(b.grid, 1, 1), block(64 * b.lanes * (1 + b.cpl), 1, 1);
  hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(fn), grid, block, params, b.lds_bytes, stream);
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "matmul_chain launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int chain_status(hipStream_t stream, uint32_t* out4) {
  unsigned char* base = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    ChainSlab* slab = chain_slab(stream, 0, false);
    if (slab) base = slab->ptr;
  }
  memset(out4, 0, 4 * sizeof(uint32_t));
  if (!base) return WQAA_OK;
  if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(out4, base, 16, hipMemcpyDeviceToHost) != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "chain_status: %s", hipGetErrorString(hipGetLastError()));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int64_t chain_trace(hipStream_t stream, uint64_t* out, int64_t max_words) {
  unsigned char* base = nullptr;
  size_t words = 0, toff = 0;
  {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    ChainSlab* slab = chain_slab(stream, 0, false);
    if (slab) { base = slab->ptr; words = slab->trace_words; toff = slab->trace_off; }
  }
  if (!base || !words || !out) return 0;
  if ((int64_t)words > max_words) words = (size_t)max_words;
  if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(out, base + toff, words * 8, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return (int64_t)words;
}

void chain_init() {
  for (int bits : {4, 2})
    for (int layout = 0; layout < 2; ++layout)
      for (int mode = 0; mode <= MD_ZR; ++mode)
      {
        chain_fn fn = pick_chain(bits, layout, mode);
        if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
  (void)hipGetLastError();
}

}  // namespace wqaa
