// wqaa_gemv.hip - host side of the GEMV family: classification, tile-config selector, launch; plus the
// activation quantiser and the decode known-answer kernel.  The GEMV kernel lives in wqaa_gemv_kernel.h,
// its member tables in wqaa_gemv_inst_*.hip.
#include "wqaa_gemv_kernel.h"

namespace wqaa {


// ------------------------------------------------------------------------------------------
// host side: tile-config selector
// ------------------------------------------------------------------------------------------
static gemv_fn pick_kernel(int kind, int layout, int at, int mode, int flags, int mb) {
  if (at == AT_F16 && (flags & FL_BF16)) {   // bfloat16 activations: plain layout
    return layout == LAYOUT_PLAIN ? pick_gemv_bf16(kind, mode, mb) : nullptr;
  }
  if (at == AT_F16) {
    if (!(flags & FL_A8) && (kind == DK_INT4 || kind == DK_INT2 || kind == DK_INT1)) return pick_gemv_f16_int(kind, layout, mode, mb);
    return pick_gemv_f16_other(kind, mode, flags, mb);
  }
  if (mode != MD_NONE) return nullptr;
  return pick_gemv_int(kind, layout, at, flags, mb);
}

struct GemvChoice {
  gemv_fn fn;
  int kind, layout, at, mode, flags, mb, R, D;
  int bits;
  int E;
  int nc, ncp, cpr;
  int grid_x, grid_y, lds, threads, variant;
  int kw, spp;                // K split across the waves of a workgroup (1 = none)
  int fp4_table;
  int a_fmt;
};

static int classify(const wqaa_matmul_desc& d, GemvChoice* c) {
  const int a = d.a_dtype;
  c->fp4_table = 0;
  c->flags = 0;
  c->a_fmt = a;
  c->layout = d.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN;
  if (a == WQAA_F16) {
    c->at = AT_F16;
  } else if (a == WQAA_BF16) {
    c->at = AT_F16;          // same machine path, bfloat16 arithmetic (FL_BF16)
    c->flags |= FL_BF16;
  } else if (a == WQAA_I8) {
    c->at = AT_I8;
  } else if (a == WQAA_I4) {
    c->at = AT_I4;
  } else if (a == WQAA_E4M3 || a == WQAA_E5M2) {
    c->at = AT_F16;
    c->flags |= FL_A8;
  } else {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: A dtype %d not supported", a);
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
      if (a == WQAA_E4M3) c->kind = DK_E4M3;
      else if (a == WQAA_E5M2) c->kind = DK_E5M2;
      else c->kind = DK_NATIVE;
      c->bits = (a == WQAA_F16 || a == WQAA_BF16) ? 16 : 8;
      if (a == WQAA_I4) { c->kind = DK_INT4; c->bits = 4; }   // two's-complement nibbles
      break;
    default: c->kind = -1;
  }
  if (a == WQAA_I4 && !(c->kind == DK_INT4 && d.w_format == WQAA_W_NATIVE) && c->kind != DK_INT2) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: int4 activations pair with int4 (native) or 2-bit weights only");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->kind < 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: weight format %d / %d bits not supported", d.w_format, d.w_bits);
    return WQAA_ERR_UNSUPPORTED;
  }
  if ((c->flags & FL_A8) && c->kind != DK_E4M3 && c->kind != DK_E5M2) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: fp8 activations need fp8 weights");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->kind == DK_E4M3 && d.strict_reference && !(c->flags & (FL_A8 | FL_BF16))) c->flags |= FL_STRICT;
  if (c->kind != DK_INT4 && c->kind != DK_INT2 && c->kind != DK_INT1) c->layout = LAYOUT_PLAIN;
  if (at_is_int(c->at) && (d.with_scaling || d.zeros_mode != WQAA_Z_NONE)) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: scale/zeros with int8 activations are not defined by the reference");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->flags & FL_BF16) {
    const bool kind_ok = c->kind == DK_INT4 || c->kind == DK_INT2 || c->kind == DK_INT1 || c->kind == DK_INT8 || c->kind == DK_NATIVE ||
                         c->kind == DK_LUT4 || c->kind == DK_E4M3;
    if (!kind_ok || c->layout != LAYOUT_PLAIN) {
      set_error(WQAA_ERR_UNSUPPORTED, "gemv: bfloat16 activations support plain-layout integer / nf4 / fp4 / e4m3 / bf16 weights");
      return WQAA_ERR_UNSUPPORTED;
    }
  }
  // zero points only act together with a scale (matmul_dequantize_impl.py:435-449)
  c->mode = !d.with_scaling ? MD_NONE
            : d.zeros_mode == WQAA_Z_ORIGINAL ? MD_ZO
            : d.zeros_mode == WQAA_Z_RESCALE  ? MD_ZR
            : d.zeros_mode == WQAA_Z_QUANTIZED ? MD_ZQ
                                               : MD_S;
  c->E = 128 / c->bits;
  if (d.K % c->E != 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: K=%d must be a multiple of %d for %d-bit weights", d.K, c->E, c->bits);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  if (d.K % g != 0 || (c->mode != MD_NONE && g % c->E != 0)) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: group_size=%d must divide K=%d and be a multiple of %d", g, d.K, c->E);
    return WQAA_ERR_UNSUPPORTED;
  }
  c->cpr = d.K / c->E;
  c->nc = (c->cpr + 63) / 64;
  return WQAA_OK;
}

// tile-config selection: batch tile MB, waves per workgroup and the grid.
// Measured on MI355X (tools/gemv_probe.hip, tools/wq_bench.cpp, profiles/): at M = 1 a small matrix
// is one latency chain - dispatch, one HBM round trip with the whole weight stream in flight, the
// per-wave decode, the store - and (R, D) = (2, 2) was the best or within 3 % of the best of the
// nine (R, D) shapes tried on 4096x4096 ... 8192x28672.  What the selector has to get right is
// residency: the activation tile lives in LDS once per workgroup, so for long K the workgroup is
// widened (up to 16 waves) until the CU holds ~32 waves, i.e. >= 64 KiB of weight loads in flight.
static int choose(const wqaa_matmul_desc& d, int m, GemvChoice* c, bool quant_in = false) {
  int st = classify(d, c);
  if (st != WQAA_OK) return st;
  const int mb = m <= 1 ? 1 : m <= 2 ? 2 : 4;
  c->mb = mb;
  c->R = 2;
  c->D = 2;
  c->variant = 0;
  c->ncp = (c->nc + c->D - 1) / c->D * c->D;
  // K within one step: the M = 1 member that keeps its activation slice in registers
  // and few enough waves per CU that the redundant per-wave reads of A stay cheap (same-box A/B, int4:
  // 1024 rows -4 %, 2048 -3 %, 4096 -2.5 %, 11008 +3 %)
  const int cus0 = device_info().ok ? device_info().cus : 256;
  if (quant_in) {
    if (c->at != AT_I8 || d.a_dtype != WQAA_I8) {
      set_error(WQAA_ERR_UNSUPPORTED, "gemv: in-kernel activation quantisation needs an int8-activation operator");
      return WQAA_ERR_UNSUPPORTED;
    }
    c->flags |= FL_AQ;
  }
  // rows with an odd number of lane chunks, integer activations: four rows x one chunk per step instead of two x two (no
  // padding chunk).  Same-process A/B, int2 x int8 (profiles/r03_ab_chunk_tile.txt): 12288 x 4096 5.95 -> 5.25 us, 22016 x 4096
  // 8.78 -> 7.42, 8640 x 3200 5.22 -> 4.53; but 4096 x 11008 (three chunks, 1024 groups of four rows: 4 waves per CU) 5.71 ->
  // 6.49 - so: one-chunk rows, or enough rows that four per wave still give every CU 8 waves.  WQAA_GEMV_TUNE=chunk=0: the (2, 2) members.
  bool chunk_tile = false;
  if (c->at == AT_I8 && c->bits < 8 && (c->nc & 1) && (c->nc == 1 || (d.N + 3) / 4 >= 8 * cus0)) {
    int cv = 1;
    (void)gemv_knob("chunk", &cv);
    chunk_tile = cv != 0 && pick_kernel(c->kind, c->layout, c->at, c->mode, c->flags, kChunkTile + mb) != nullptr;
  }
  if (chunk_tile) {
    c->R = 4;
    c->D = 1;
  }
  c->ncp = (c->nc + c->D - 1) / c->D * c->D;
  // the register-resident member only where its activation slice fits the register file (gemv_direct_fits, wqaa_gemv_kernel.h:
  // the members that spilled are not built; same-process A/B in profiles/r03_ab_direct_fit.txt)
  const bool slice_fits = mb * c->E * (c->at == AT_F16 ? 2 : 1) <= 128 && !(c->kind == DK_LUT4 && mb == 2 && (c->flags & FL_BF16));
  int areg_knob = 1;                                    // WQAA_GEMV_TUNE=areg=0: the LDS-staged members (A/B and test aid)
  (void)gemv_knob("areg", &areg_knob);
  const bool direct = mb <= 2 && m == mb && !(c->flags & (FL_A8 | FL_AQ)) && c->at != AT_I4 && c->ncp == c->D && (d.N + c->R - 1) / c->R <= 10 * cus0 &&
                      slice_fits && areg_knob != 0;
  // small matrices: one row per wave doubles the waves in flight (same-box A/B: 1024 x 1024 2.87 -> 2.45 us,
  // 2048 x 4096 equal, 4096 x 4096 4.18 -> 4.43 us)
  const bool r1 = direct && mb == 1 && (d.N + 1) / 2 < 3 * cus0 && !chunk_tile;
  if (r1) c->R = 1;
  int code = direct ? (mb == 2 ? kDirectTile + 2 : r1 ? kDirectTile + 1 : kDirectTile) : mb;
  if (chunk_tile) code = (direct ? kChunkDirect : kChunkTile) + mb;
  c->fn = pick_kernel(c->kind, c->layout, c->at, c->mode, c->flags, code);
  if (!c->fn) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: no kernel for kind=%d layout=%d at=%d mode=%d flags=%d", c->kind,
              c->layout, c->at, c->mode, c->flags);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int cus = device_info().ok ? device_info().cus : 256;
  const long kpad = (long)c->ncp * 64 * c->E;
  c->lds = (int)(mb * kpad * (c->at == AT_F16 ? 2 : 1));
  if (direct) { c->lds = 0; c->variant = 1; }
  if (c->flags & FL_AQ) c->lds += 256;   // per-wave row maxima of the in-kernel quantiser
  if (c->lds > 160 * 1024) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemv: activation tile %d B exceeds LDS", c->lds);
    return WQAA_ERR_UNSUPPORTED;
  }
  int blocks_per_cu = 160 * 1024 / (c->lds > 0 ? c->lds : 1);
  if (blocks_per_cu > 8) blocks_per_cu = 8;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  int nw = 4;
  while (nw < 16 && blocks_per_cu * nw < 32) nw *= 2;
  // ... but never so wide that some CUs get no workgroup at all (N = 4096, K = 11008, M = 2: 16 waves
  // gave 128 workgroups for 256 CUs - 14.8 us; 8 waves, one workgroup per CU - see DESIGN 3.4)
  {
    const int n_rg0 = (d.N + c->R - 1) / c->R;
    while (nw > 4 && (n_rg0 + nw - 1) / nw < cus) nw /= 2;
  }
  const int n_rg = (d.N + c->R - 1) / c->R;
  // Few-row shards (N / 8 slices of a column-parallel layer: 1024 x 28672, 1280 x 8192): one wave per row group leaves
  // most of the chip without a wave.  Split K across kw waves of a workgroup until ~8 waves per CU are busy; the parts
  // meet in LDS in a fixed order (bit-stable; integer members bit-exact).  Not for the register-resident members (one
  // step) nor the in-kernel quantiser.
  c->kw = 1;
  {
    const int nsteps = c->ncp / c->D;
    const bool can = !direct && !chunk_tile && !(c->flags & FL_AQ) && m <= mb && mb <= 2 && nsteps >= 2 &&
                     pick_kernel(c->kind, c->layout, c->at, c->mode, c->flags, kSplitTile + mb) != nullptr;
    int kw = 1;
    if (can && n_rg < 4 * cus) {
      kw = (8 * cus + n_rg - 1) / n_rg;              // waves per row group that bring the chip to ~8 waves per CU
      if (kw > 8) kw = 8;
      if (kw > nsteps) kw = nsteps;
      while (kw > 1 && ((nsteps + kw - 1) / kw) * (kw - 1) >= nsteps) --kw;   // every part gets at least one step
    }
    if (can && d.k_split_hint > 1) kw = d.k_split_hint;       // the caller's MatmulConfigWithSplitK.k_split
    { int v; if (gemv_knob("kw", &v) && can && v > 0) kw = v; }   // tuning aid
    if (kw > nsteps) kw = nsteps;
    if (kw > 16) kw = 16;
    while (kw > 1 && ((nsteps + kw - 1) / kw) * (kw - 1) >= nsteps) --kw;
    if (kw > 1 && c->lds + 2 * 16 * c->R * mb * 4 > 160 * 1024) kw = 1;   // no room for the parts' partial sums
    if (kw > 1) {
      int slots = nw / kw;                            // keep the workgroup near the width chosen above
      if (slots < 1) slots = 1;
      while (slots > 1 && (n_rg + slots - 1) / slots < cus) slots /= 2;
      if (slots * kw > 16) slots = 16 / kw;
      if (slots < 1) slots = 1;
      nw = slots * kw;
      c->kw = kw;
      c->fn = pick_kernel(c->kind, c->layout, c->at, c->mode, c->flags, kSplitTile + mb);   // the K-split twin
      c->spp = (nsteps + kw - 1) / kw;
      c->lds += 2 * nw * c->R * mb * 4;               // the parts' partial sums, double buffered
    }
  }
  if (blocks_per_cu * nw > 32) blocks_per_cu = 32 / nw;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  c->threads = nw * 64;
  const int rg_per_block = nw / c->kw;
  int blocks = (n_rg + rg_per_block - 1) / rg_per_block;
  const int cap = cus * blocks_per_cu;
  if (blocks > cap && !gemv_uncapped()) blocks = cap;
  if (blocks < 1) blocks = 1;
  { int v; if (gemv_knob("grid", &v) && v > 0) blocks = v; }   // tuning aid
  if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;   // whole XCD rounds: keeps the block swizzle on
  c->grid_x = blocks;
  c->grid_y = (m + mb - 1) / mb;
  return WQAA_OK;
}

static void fill_args(const wqaa_matmul_desc& d, const GemvChoice& c, const void* A, const void* B,
                      const void* LUT, const void* Scale, const void* Zeros, const void* Bias, void* C,
                      int m, GemvArgs* a) {
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  a->A = A; a->B = B; a->lut = LUT; a->scale = Scale; a->zeros = Zeros; a->bias = Bias; a->C = C;
  a->m = m; a->N = d.N; a->K = d.K;
  a->kg = d.K / g;
  a->g = g;
  a->g_log2 = ilog2_exact(g);
  {
    const int dq = g / c.E > 0 ? g / c.E : 1;      // lane chunks per group (mode != NONE => g % E == 0)
    a->gq_shift = ilog2_exact(dq);
    a->gq_magic = a->gq_shift >= 0 ? 0u : (uint32_t)(((1ull << 32) + dq - 1) / dq);
  }
  a->nc = c.nc;
  a->ncp = c.ncp;
  a->cpr = c.cpr;
  a->row_bytes = (long)d.K * c.bits / 8;
  a->has_bias = d.with_bias;
  a->out_dtype = d.out_dtype;
  // "uint8" weights under strict_reference: the TE graph reads the int8 storage buffer with `.astype(A_dtype)`
  // (matmul_dequantize_impl.py:404-406), i.e. SIGNED bytes - pinned by tests/golden/te_golden.npz (f16_uint8_scale)
  a->is_signed = d.w_format == WQAA_W_INT || (d.w_format == WQAA_W_UINT && d.w_bits == 8 && d.strict_reference);
  // int4 activations: 4-bit weights are native two's complement, 2-bit weights are zero-extended
  // (matmul_dequantize_mma.py:742-749)
  if (d.a_dtype == WQAA_I4) a->is_signed = c.kind == DK_INT4;
  a->fp4_table = c.fp4_table;
  a->a_fmt = c.a_fmt;
  a->zq_row_bytes = d.N * (c.bits < 8 ? c.bits : 8) / 8;
  a->epi_row = nullptr;
  a->epi_tensor = 1.f;
  a->kw = c.kw;
  a->kw_magic = (65536u + (uint32_t)c.kw - 1u) / (uint32_t)c.kw;
  a->spp = c.kw > 1 ? c.spp : 0;
  {
    const int rg_per_block = (c.threads / 64) / c.kw;
    a->n_rgb = ((d.N + c.R - 1) / c.R + rg_per_block - 1) / rg_per_block;
  }
}

int gemv_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan) {
  if (gemvx_eligible(d, m)) return gemvx_plan(d, m, plan);
  GemvChoice c;
  int st = choose(d, m, &c);
  if (st != WQAA_OK) return st;
  if (plan) {
    plan->kernel_family = 1;
    plan->block_m = c.mb;
    plan->block_n = c.R * (c.threads / 64);
    plan->block_k = 64 * c.E * c.D;
    plan->threads = c.threads;
    plan->grid = c.grid_x * c.grid_y;
    plan->rows_per_wave = c.R;
    plan->batch_tile = c.mb;
    plan->pipeline_depth = c.D;
    plan->split_k = c.kw;
    plan->lds_bytes = c.lds;
    char wd[24];
    short_wdtype(d, wd, sizeof(wd));
    char ks[8] = "";
    if (c.kw > 1) snprintf(ks, sizeof(ks), "k%d", c.kw);
    snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_gemv_b%dr%dd%d%s%s", m, d.N, d.K,
             short_dtype(d.a_dtype), wd, c.mb, c.R, c.D, ks, c.variant ? "_areg" : "");
  }
  return WQAA_OK;
}

static int gemv_dispatch(const GemvChoice& c, GemvGroupArgs& ga, int grid_x, int count, hipStream_t stream, hipEvent_t start,
                         hipEvent_t stop) {
  void* params[] = {&ga};
  dim3 grid(grid_x, c.grid_y, count), block(c.threads, 1, 1);
  hipError_t e;
  if (start || stop) {
    e = hipExtLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream, start, stop, 0);
  } else {
    e = hipLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream);
  }
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "gemv launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int gemv_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi) {
  if (!epi && gemvx_eligible(d, m)) return gemvx_launch(d, A, B, Scale, Zeros, Bias, C, m, stream, start, stop);
  if (epi && (epi->flags & (WQAA_EPI_ADD_RESIDUAL | WQAA_EPI_RMSNORM_INPUT))) {
    // the float16 path's residual add / norm in front exist in the exact-product family only (whatever strict_reference says:
    // the caller asked for a fusion the per-element-rounding definition does not have)
    if (!gemvx_covers(d, m) || d.out_dtype != WQAA_F16) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_ex: residual add / RMSNorm input need float16 activations, 1/2/4-bit integer weights, "
                                      "float16 output and m <= 2 (got m=%d)", m);
      return WQAA_ERR_UNSUPPORTED;
    }
    return gemvx_launch(d, A, B, Scale, Zeros, Bias, C, m, stream, start, stop, epi);
  }
  GemvChoice c;
  {
    static thread_local ChoiceMemo<GemvChoice> memo;
    const int q = (epi && (epi->flags & WQAA_EPI_QUANTIZE_INPUT)) ? 1 : 0;
    if (const GemvChoice* hit = memo.find(d, m, q)) {
      c = *hit;
    } else {
      int st = choose(d, m, &c, q != 0);
      if (st != WQAA_OK) return st;
      memo.put(d, m, q, c);
    }
  }
  GemvGroupArgs ga;
  GemvArgs& a = ga.p[0];
  fill_args(d, c, A, B, LUT, Scale, Zeros, Bias, C, m, &a);
  if (epi) {
    if (!at_is_int(c.at) || d.out_dtype != WQAA_F16) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_ex: the fused epilogue needs int8 activations and float16 output");
      return WQAA_ERR_UNSUPPORTED;
    }
    a.epi_row = epi->row_scale;
    a.epi_tensor = epi->tensor_scale;
  }
  return gemv_dispatch(c, ga, c.grid_x, 1, stream, start, stop);
}

// ---- a group of independent operators in one launch (wqaa_matmul_group): tile configuration of the MERGED operator
// (N = the sum of the members' rows), every member takes gridDim.x x gridDim.y workgroups of it (blockIdx.z = member) ----
static int gemv_group_choose(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, GemvChoice* c, int* grid_x,
                             bool quant_in = false) {
  {
    static thread_local ChoiceMemo<GemvChoice> memo;
    const int q = 16 + count + (quant_in ? 64 : 0);
    if (const GemvChoice* hit = memo.find(merged, m, q)) {
      *c = *hit;
    } else {
      int st = choose(merged, m, c, quant_in);
      if (st != WQAA_OK) return st;
      memo.put(merged, m, q, *c);
    }
  }
  const int rg_per_block = (c->threads / 64) / c->kw;
  int need = 1;
  for (int i = 0; i < count; ++i) {
    const int blocks = ((Ns[i] + c->R - 1) / c->R + rg_per_block - 1) / rg_per_block;
    if (blocks > need) need = blocks;
  }
  int gx = need;                         // one row-group block per workgroup, sized by the largest member (see gemvx_group_choose)
  { int v; if (gemv_knob("group_grid", &v) && v > 0 && v < gx) gx = v; }    // tuning aid
  if (gx >= 8) gx = (gx + 7) / 8 * 8;
  *grid_x = gx;
  return WQAA_OK;
}

bool gemv_group_eligible(const wqaa_matmul_desc& merged, const wqaa_matmul_desc* const* descs, int count, int m, bool with_epilogue, bool quant_in) {
  if (count < 1 || count > kGemvGroupMax || m < 1 || m > 2) return false;
  GemvChoice c;
  if (choose(merged, m, &c, quant_in) != WQAA_OK) return false;      // (the caller restores the error side channel)
  // bit-for-bit with the single calls: every member alone must get the merged operator's member variant, K split and depth
  // (the summation order of a row; see gemvx_group_eligible).  Integer accumulation is order-free: nothing to compare there.
  if (!at_is_int(c.at)) {
    for (int i = 0; descs && i < count; ++i) {
      GemvChoice ci;
      if (choose(*descs[i], m, &ci, quant_in) != WQAA_OK) return false;
      if (ci.kw != c.kw || ci.D != c.D) return false;
    }
  }
  return !with_epilogue || (at_is_int(c.at) && merged.out_dtype == WQAA_F16);
}

int gemv_group_plan(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, wqaa_plan* plan) {
  GemvChoice c;
  int gx = 0;
  int st = gemv_group_choose(merged, Ns, count, m, &c, &gx);
  if (st != WQAA_OK) return st;
  st = gemv_plan(merged, m, plan);
  if (st == WQAA_OK && plan) {
    plan->grid = gx * c.grid_y * count;
    char tail[16];
    snprintf(tail, sizeof(tail), "_x%d", count);
    strncat(plan->name, tail, sizeof(plan->name) - strlen(plan->name) - 1);
  }
  return st;
}

int gemv_group_launch(const wqaa_matmul_desc& merged, const wqaa_group_item* items, int count, int m, hipStream_t stream,
                      const wqaa_epilogue* const* epis) {
  int Ns[kGemvGroupMax];
  for (int i = 0; i < count; ++i) Ns[i] = items[i].desc->N;
  GemvChoice c;
  int gx = 0;
  const bool quant_in = epis && (epis[0]->flags & WQAA_EPI_QUANTIZE_INPUT);
  int st = gemv_group_choose(merged, Ns, count, m, &c, &gx, quant_in);
  if (st != WQAA_OK) return st;
  if (epis && (!at_is_int(c.at) || merged.out_dtype != WQAA_F16)) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_group_ex: the fused epilogue needs int8 activations and float16 output");
    return WQAA_ERR_UNSUPPORTED;
  }
  GemvGroupArgs ga;
  for (int i = 0; i < count; ++i) {
    fill_args(*items[i].desc, c, items[i].A, items[i].B, items[i].LUT, items[i].Scale, items[i].Zeros, items[i].Bias, items[i].C, m,
              &ga.p[i]);
    if (epis) {
      ga.p[i].epi_row = epis[i]->row_scale;
      ga.p[i].epi_tensor = epis[i]->tensor_scale;
    }
  }
  return gemv_dispatch(c, ga, gx, count, stream, nullptr, nullptr);
}

void gemv_init() {
  gemvx_init();
  // raise the dynamic-LDS ceiling of every family member to the full 160 KiB
  const int kinds[] = {DK_INT4, DK_INT2, DK_INT1, DK_INT8, DK_LUT4, DK_E4M3, DK_E5M2, DK_NATIVE};
  for (int kind : kinds)
    for (int layout = 0; layout < 2; ++layout)
      for (int at : {(int)AT_F16, (int)AT_I8, (int)AT_I4})
        for (int mode = 0; mode <= MD_ZQ; ++mode)
          for (int flags : {0, 1, 2, 3, (int)FL_BF16, (int)FL_AQ})
            for (int mb : kBatchTiles) {
              gemv_fn fn = pick_kernel(kind, layout, at, mode, flags, mb);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);

            }
  (void)hipGetLastError();   // a refused attribute must not linger as this thread's "last error"
}


// ------------------------------------------------------------------------------------------
// per-row absmax int8 quantiser - the pre-op of BitNet-style callers (integration/BitNet/
// utils_quant.py:161-168): s = (1 / clamp(max|x|, 1e-5)) * 127 (see act_quant_scale), q = clamp(round(x * s), -128, 127).
// One workgroup per row, 16-byte loads, wave DPP max + LDS across waves.  HBM-bound: 3 B per element.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wq_act_quant_kernel(const half_t* __restrict__ X, int K, int8_t* __restrict__ Q,
                                                           float* __restrict__ S) {
  __shared__ float wmax[4];
  const long row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(X + row * K);
  const int nvec = K / 8;
  float mx = 0.f;
  for (int i = threadIdx.x; i < nvec; i += 256) {
    const u32x4 v = xr[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half2_t h = as_h2(v[e]);
      mx = fmaxf(mx, fmaxf(fabsf((float)h[0]), fabsf((float)h[1])));
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  const float s = act_quant_scale(mx);
  if (threadIdx.x == 0) S[row] = s;
  u32x2* qr = reinterpret_cast<u32x2*>(Q + row * K);
  for (int i = threadIdx.x; i < nvec; i += 256) {
    const u32x4 v = xr[i];
    uint32_t out[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half2_t h = as_h2(v[e >> 1]);
      float q = rintf((float)h[e & 1] * s);           // torch.round: half to even
      q = fminf(fmaxf(q, -128.f), 127.f);
      out[e >> 2] |= ((uint32_t)(int)q & 0xFFu) << (8 * (e & 3));
    }
    qr[i] = u32x2{out[0], out[1]};
  }
}

int act_quant_launch(const void* X, int64_t rows, int K, void* Q, float* S, hipStream_t stream) {
  if (!X || !Q || !S || rows < 0 || K <= 0 || K % 8 != 0) {
    set_error(WQAA_ERR_BAD_DESC, "act_quant: bad arguments (K=%d must be a multiple of 8)", K);
    return WQAA_ERR_BAD_DESC;
  }
  if (rows == 0) return WQAA_OK;
  hipLaunchKernelGGL(wq_act_quant_kernel, dim3((unsigned)rows), dim3(256), 0, stream, reinterpret_cast<const half_t*>(X), K,
                     reinterpret_cast<int8_t*>(Q), S);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "act_quant launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

// ------------------------------------------------------------------------------------------
// decode known-answer kernel (HIP twin of testing/cpp/lop3_type_conversion/*.cu): decode packed
// words with the exact routines above and scatter the values back to source order.
// ------------------------------------------------------------------------------------------
template <int KIND, int LAYOUT, int AT>
__global__ void debug_decode_kernel(const uint32_t* packed, long nwords, int is_signed, int strict,
                                    int fp4_table, const half_t* lut_p, void* out) {
  using T = KindTraits<KIND, AT>;
  constexpr int EPW = T::EPW;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  const uint32_t w = packed[i];
  if constexpr (AT == AT_F16) {
    half2_t q[EPW / 2 > 0 ? EPW / 2 : 1];
    const half_t zf = (is_signed && T::SUBBYTE) ? (half_t)(float)(1 << (T::BITS - 1)) : (half_t)0.0f;
    if constexpr (KIND == DK_INT4 || KIND == DK_INT2 || KIND == DK_INT1) {
      uint32_t magic[8];
      make_magic(magic);
      F16Unpack<T::BITS>::run((KIND == DK_INT1 && is_signed) ? ~w : w, zf, magic, q);
    } else if constexpr (KIND == DK_LUT4) {
      Lut16 lut = fp4_table ? make_fp4_lut() : make_lut16(lut_p);
      lut16_word(lut, w, q);
    } else if constexpr (KIND == DK_INT8) {
      half2_t t[2];
      if (is_signed) unpack8_f16<true>(w, zf, t); else unpack8_f16<false>(w, zf, t);
      q[0] = t[0]; q[1] = t[1];
    } else if constexpr (KIND == DK_E4M3) {
      half2_t t[2];
      if (strict) unpack_e4m3_f16<true>(w, t); else unpack_e4m3_f16<false>(w, t);
      q[0] = t[0]; q[1] = t[1];
    } else if constexpr (KIND == DK_E5M2) {
      half2_t t[2];
      unpack_e5m2_f16(w, t);
      q[0] = t[0]; q[1] = t[1];
    } else {
      q[0] = as_h2(w);
    }
    half_t* o = reinterpret_cast<half_t*>(out) + i * EPW;
#pragma unroll
    for (int x = 0; x < EPW; ++x) {
      const int src = T::SUBBYTE ? src_of_field(T::BITS, T::S, LAYOUT, T::field_of_slot(x)) : x;
      o[src] = q[x / 2][x & 1];
    }
  } else {
    int8_t* o = reinterpret_cast<int8_t*>(out) + i * EPW;
    if constexpr (T::SUBBYTE) {
      uint32_t q[I8Unpack<T::BITS>::NQUAD];
      I8Unpack<T::BITS>::run((T::BITS == 1 && is_signed) ? ~w : w, q);
      const uint32_t zp4 = is_signed ? (uint32_t)(1u << (T::BITS - 1)) * 0x01010101u : 0u;
#pragma unroll
      for (int x = 0; x < EPW; ++x) {
        const uint32_t v = zp4 ? sub_bytes(q[x / 4], zp4) : q[x / 4];
        const int src = src_of_field(T::BITS, T::S, LAYOUT, T::field_of_slot(x));
        o[src] = (int8_t)(v >> (8 * (x & 3)));
      }
    } else {
#pragma unroll
      for (int x = 0; x < 4; ++x) o[x] = (int8_t)(w >> (8 * x));
    }
  }
}

int debug_decode_launch(const void* packed, int64_t nwords, int w_format, int bits, int layout,
                        int a_dtype, int strict, const void* lut, void* out, hipStream_t stream) {
  wqaa_matmul_desc d = {};
  d.a_dtype = a_dtype; d.w_format = w_format; d.w_bits = bits; d.w_layout = layout;
  d.K = 128; d.N = 1; d.group_size = -1;
  GemvChoice c;
  int st = classify(d, &c);
  if (st != WQAA_OK) return st;
  const int is_signed = w_format == WQAA_W_INT;
  const dim3 block(256), grid((unsigned)((nwords + 255) / 256));
  const uint32_t* p = reinterpret_cast<const uint32_t*>(packed);
  const half_t* lp = reinterpret_cast<const half_t*>(lut);
  const int str = strict && w_format == WQAA_W_E4M3;
#define WQAA_DBG(K, L, A) \
  if (c.kind == K && c.layout == L && c.at == A) { \
    hipLaunchKernelGGL((debug_decode_kernel<K, L, A>), grid, block, 0, stream, p, (long)nwords, is_signed, str, c.fp4_table, lp, out); \
    return WQAA_OK; }
  WQAA_DBG(DK_INT4, LAYOUT_LOP3, AT_F16) WQAA_DBG(DK_INT4, LAYOUT_PLAIN, AT_F16)
  WQAA_DBG(DK_INT2, LAYOUT_LOP3, AT_F16) WQAA_DBG(DK_INT2, LAYOUT_PLAIN, AT_F16)
  WQAA_DBG(DK_INT1, LAYOUT_LOP3, AT_F16) WQAA_DBG(DK_INT1, LAYOUT_PLAIN, AT_F16)
  WQAA_DBG(DK_INT8, LAYOUT_PLAIN, AT_F16) WQAA_DBG(DK_LUT4, LAYOUT_PLAIN, AT_F16)
  WQAA_DBG(DK_E4M3, LAYOUT_PLAIN, AT_F16) WQAA_DBG(DK_E5M2, LAYOUT_PLAIN, AT_F16)
  WQAA_DBG(DK_INT4, LAYOUT_LOP3, AT_I8) WQAA_DBG(DK_INT4, LAYOUT_PLAIN, AT_I8)
  WQAA_DBG(DK_INT2, LAYOUT_LOP3, AT_I8) WQAA_DBG(DK_INT2, LAYOUT_PLAIN, AT_I8)
  WQAA_DBG(DK_INT1, LAYOUT_LOP3, AT_I8) WQAA_DBG(DK_INT1, LAYOUT_PLAIN, AT_I8)
#undef WQAA_DBG
  set_error(WQAA_ERR_UNSUPPORTED, "debug_decode: unsupported combination");
  return WQAA_ERR_UNSUPPORTED;
}

}  // namespace wqaa
