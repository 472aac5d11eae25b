// wqaa_gemvx.hip - host side of the exact-product GEMV members (wqaa_gemvx_kernel.h): eligibility, tile-config
// selection (rows per wave, K split across the waves of a workgroup, workgroup width, grid) and launch.
#include "wqaa_gemvx_kernel.h"

namespace wqaa {

struct GemvxChoice {
  gemvx_fn fn;
  int bits, layout, mode, mb, R, D, kw, nw;
  int E, cpr, nc, nsteps, n_rgb;
  int grid, lds, areg;
};

// sub-byte integer weights x float16 activations, M <= 2, and the caller did not ask for the TE definition's
// per-element rounding (strict_reference).  WQAA_GEMV_TUNE=exact=0 disables the family (A/B aid).
// what the family can compute at all (the fused pre/post ops of wqaa_matmul_ex exist only here, so they ask this)
bool gemvx_covers(const wqaa_matmul_desc& d, int m) {
  if (m < 1 || m > 2 || d.a_dtype != WQAA_F16) return false;
  if (d.w_format != WQAA_W_UINT && d.w_format != WQAA_W_INT) return false;
  if (d.w_bits != 4 && d.w_bits != 2 && d.w_bits != 1) return false;
  const int E = 128 / d.w_bits;
  if (d.K % E != 0) return false;
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  if (d.K % g != 0 || (d.with_scaling && g % E != 0)) return false;
  return true;
}

bool gemvx_eligible(const wqaa_matmul_desc& d, int m) {
  if (d.strict_reference || !gemvx_covers(d, m)) return false;
  // Long K with enough rows to fill the chip unsplit, two activation rows: every workgroup stages both activation rows
  // and their chunk sums before its first dot, which the rounding members (wider workgroups, no sums) do cheaper
  // (same-call A/B, int4 g128, exact vs rounding member, profiles/r02_ab_gemvx_longk.txt: M=2 4096x11008 11.2 vs 10.2 us).
  // At M = 1 the exact members keep long K too since their workgroups take 16 rows there (gemvx_choose; profiles/
  // r02_ab_knobs_longk.txt: 5120x13824 13.0 vs 14.8 us, 8192x28672 24.7 vs 26.7, 4096x14336 9.9-10.2 vs 9.6-10.2).
  // Either numerics meets the contract when strict_reference = 0: the faster member is taken.
  // the switch is a plan-time one like every tuning variable (ChoiceMemo): re-read when wqaa_select bumps the epoch - ONE getenv per
  // epoch, not per call (the `forced` test used to read the environment on every eligibility question)
  static thread_local unsigned seen_epoch = 0;
  static thread_local bool enabled = true, forced = false;
  const unsigned ep = g_plan_epoch.load(std::memory_order_relaxed);
  if (ep != seen_epoch) {
    int v = 1;
    const bool have = gemv_knob("exact", &v);
    enabled = !(have && v == 0);
    forced = have && v == 2;                                  // WQAA_GEMV_TUNE=exact=2: A/B aid, ignores the fences
    seen_epoch = ep;
  }
  {
    const int cus = device_info().ok ? device_info().cus : 256;
    if (!forced && d.K > 8192 && (d.N + 1) / 2 >= 8 * cus && m > 1) return false;
    // two activation rows: twice the LDS reads and dots per weight word - the exact member only wins on many-row matrices
    // (same-call, int4 g128: 11008x4096 8.8 vs 9.4 us; 4096^2 5.4 vs 5.1, 4096x11008 11.2 vs 10.2)
    if (!forced && m == 2 && d.N < 8192) return false;
  }
  return enabled;
}

// Tile-config selection.  Many small workgroups (the hardware dispatcher balances them; a persistent workgroup per CU
// measured 15-40 % slower): a workgroup works on `slots` row groups at a time, each by `kw` waves that split K.
// For R in {2, 1}: the K split that brings the waves per CU towards 16 (never more parts than steps, and only splits
// that leave the parts evenly loaded); the candidate with more busy waves wins, two rows per wave (half the LDS reads
// per weight byte) on a tie.
static void gemvx_candidate(int N, int nsteps, int cus, int R, int* kw, double* score) {
  const int n_rg = (N + R - 1) / R;
  const double base = (double)n_rg / cus;            // waves per CU without a split
  int want = 1;
  while (want < 8 && base * want < 16.0) ++want;
  if (want > nsteps) want = nsteps;
  int best = 1;
  for (int k = want; k >= 1; --k) {
    const double eff = (double)nsteps / (k * ((nsteps + k - 1) / k));
    if (eff >= 0.85) { best = k; break; }
  }
  *kw = best;
  const double eff = (double)nsteps / (best * ((nsteps + best - 1) / best));
  double waves = base * best;
  if (waves > 16.0) waves = 16.0;
  *score = waves * eff;
}

// pro: 0 plain members, 1 residual add (WQAA_EPI_ADD_RESIDUAL), 2 gate / up pair (wqaa_matmul_gate_up: d.N = the rows the
// launch streams, 2 x the projections' N; two rows per wave by construction), 3 RMSNorm in front (WQAA_EPI_RMSNORM_INPUT), 4 = 3 + 2
static int gemvx_choose(const wqaa_matmul_desc& d, int m, GemvxChoice* c, int pro = 0, int force_kw = 0) {
  c->bits = d.w_bits;
  c->layout = d.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN;
  c->mode = !d.with_scaling ? MD_NONE
            : d.zeros_mode == WQAA_Z_ORIGINAL ? MD_ZO
            : d.zeros_mode == WQAA_Z_RESCALE  ? MD_ZR
            : d.zeros_mode == WQAA_Z_QUANTIZED ? MD_ZQ
                                               : MD_S;
  c->mb = m;
  c->E = 128 / c->bits;
  c->cpr = d.K / c->E;
  c->nc = (c->cpr + 63) / 64;
  c->D = 2;
  c->nsteps = (c->nc + c->D - 1) / c->D;
  const int cus = device_info().ok ? device_info().cus : 256;
  int k2, k1;
  double sc2, sc1;
  gemvx_candidate(d.N, c->nsteps, cus, 2, &k2, &sc2);
  gemvx_candidate(d.N, c->nsteps, cus, 1, &k1, &sc1);
  c->R = sc2 >= 0.9 * sc1 ? 2 : 1;
  if (pro == 2 || pro == 4) c->R = 2;
  int kw = c->R == 2 ? k2 : k1;
  if (d.k_split_hint > 1) kw = d.k_split_hint;                                          // the caller's k_split
  { int v; if (gemv_knob("kw", &v)) kw = v > 0 ? v : 1; }                              // tuning aid
  if (force_kw > 0) kw = force_kw;                                                      // a pair sums a row as its projection alone does
  if (kw > c->nsteps) kw = c->nsteps;
  if (kw > 16) kw = 16;
  c->kw = kw;
  const int n_rg = (d.N + c->R - 1) / c->R;
  // workgroup: ~8 waves (a multiple of kw); the activation tile is staged once per workgroup.  Same-call A/B, int4 g128
  // (tools/ab_gemvx_quick.sh): 8 waves beat 4 on 4096^2 (4.27 vs 4.63 us), 11008x4096 (6.92 vs 7.56) and 4096x11008
  // (8.5 vs 10.4); on streams far beyond the caches 4 waves win (28672x8192: 24.7 vs 27.4 us)
  const long wbytes = (long)d.N * d.K * c->bits / 8;
  int slots = (wbytes >= (48l << 20) ? 4 : 8) / kw;
  if (slots < 1) slots = 1;
  // ... but never so wide that CUs are left without a workgroup
  while (slots > 1 && (n_rg + slots - 1) / slots < cus) slots /= 2;
  // Long K (three lane-chunk steps or more): every workgroup stages the whole activation row and its chunk sums, 4 B per
  // B of weights of ONE row - half the weight bytes of an 8-row workgroup.  Twice the rows per workgroup (up to 16 waves)
  // wherever the workgroups still fill the chip in whole rounds (same-process A/B, tools/ab_knobs.py, profiles/
  // r02_ab_knobs_longk.txt): 4096x11008 7.7 -> 7.4 us, 1024x28672 8.6 -> 7.65, 8192x28672 29.2 -> 24.7 (now ahead of the
  // rounding member's 26.7), 14336x12288 22.0 -> 20.6; 5120x13824 would leave 320 workgroups for 256 CUs (13.0 -> 16.4):
  // not taken there.
  if (c->nsteps >= 3) {
    while (slots * 2 * kw <= 16) {
      const int nb = (n_rg + slots * 2 - 1) / (slots * 2);
      const int rounds = (nb + cus - 1) / cus;
      // one or two rounds of workgroups must be whole ones (> 5 % of a round idle otherwise); three and more even out
      if (nb < cus || (nb < 3 * cus && (long)rounds * cus * 100 > (long)nb * 105)) break;
      slots *= 2;
    }
  }
  if (slots * kw > 16) slots = 16 / kw;
  if (slots < 1) slots = 1;
  c->nw = slots * kw;
  c->n_rgb = (n_rg + slots - 1) / slots;
  const long ncp = (long)c->nsteps * c->D;
  c->lds = (int)(c->mb * ncp * 64 * (c->E * 2 + 16)) + 2 * c->nw * c->R * c->mb * 4 + 64;
  if (c->lds > 160 * 1024) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemvx: activation tile %d B exceeds LDS", c->lds);
    return WQAA_ERR_UNSUPPORTED;
  }
  int blocks_per_cu = 160 * 1024 / c->lds;
  if (blocks_per_cu > 32 / c->nw) blocks_per_cu = 32 / c->nw;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  int blocks = c->n_rgb;
  // more row-group blocks than workgroups the chip holds at once: cap the grid and let the workgroups take several
  // blocks of their XCD's eighth (28672x4096 2048 vs 3584 workgroups: 13.8 vs 14.3 us; 28672x8192 1792 vs 3584: 24.9
  // vs 27.0) - unless the second round would be a short one, where the hardware dispatcher balances better than a
  // static assignment (22016x4096, 1376 blocks: 1024 workgroups 11.5 us, 1376 10.7; tools/ab_grid.py, ab_cap.py)
  if (blocks > cus * blocks_per_cu + cus * blocks_per_cu / 2 && !gemv_uncapped()) blocks = cus * blocks_per_cu;
  if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;       // whole XCD rounds keep the block swizzle on
  { int v; if (gemv_knob("grid", &v)) blocks = v > 0 ? v : 1; }
  c->grid = blocks;
  // register-resident activations (4-bit LOP3, M = 1, K within one step, no K split): no LDS tile, no barrier.
  // WQAA_GEMV_TUNE=areg=0/1 forces it off/on (A/B aid).
  c->areg = 0;
  if (c->bits == 4 && c->layout == LAYOUT_LOP3 && c->mb == 1 && c->nsteps == 1 && c->kw == 1) {
    // same-call A/B against the LDS-staged member (profiles/r02_ab_gemvx_areg.txt): 1024 rows 2.81 -> 2.62 us, 2048 3.27 ->
    // 3.22, 28672 15.5 -> 14.8; 4096 rows 4.07 -> 4.38, 11008 6.64 -> 7.08, 12288 6.93 -> 7.37 (every wave re-reads the
    // activation row through the texture path and sums it itself) - so: few rows, or so many that the barrier of
    // tens of workgroups per CU costs more than the re-reads
    c->areg = (d.N <= 2048 || d.N >= 24576) ? 1 : 0;
    { int v; if (gemv_knob("areg", &v)) c->areg = v != 0; }
  }
  if (pro) c->areg = 0;                                  // the fused post ops come with the LDS-staged members
  if (c->areg) c->lds = 64;
  const int rd = c->R * 10 + c->D + (c->areg ? 1 : 0);
  if (pro >= 3) {
    // the norm takes sum x^2 over items held in registers: the whole activation tile must fit the items a workgroup loads ahead
    const long items = (long)c->mb * ncp * 256;
    const int nai = c->R == 1 ? 3 : 2;                     // GemvxPolicy::NAI of the NORM members
    if (items > (long)nai * c->nw * 64) {
      set_error(WQAA_ERR_UNSUPPORTED, "gemvx: RMSNorm input needs the activation rows within %d items per thread (K = %d, %d threads)", nai,
                d.K, c->nw * 64);
      return WQAA_ERR_UNSUPPORTED;
    }
    const int prd = pro * 1000 + rd;
    c->fn = c->bits == 4 ? pick_gemvx_norm4(c->layout, c->mode, c->mb, prd)
            : c->bits == 2 ? pick_gemvx_norm2(c->layout, c->mode, c->mb, prd)
                           : pick_gemvx_norm1(c->layout, c->mode, c->mb, prd);
  } else if (pro) {
    const int prd = pro == 2 ? 1000 + rd : rd;
    c->fn = c->bits == 4 ? pick_gemvx_pro4(c->layout, c->mode, c->mb, prd)
            : c->bits == 2 ? pick_gemvx_pro2(c->layout, c->mode, c->mb, prd)
                           : pick_gemvx_pro1(c->layout, c->mode, c->mb, prd);
  } else
  c->fn = c->bits == 4 ? pick_gemvx_int4(c->layout, c->mode, c->mb, rd)
          : c->bits == 2 ? pick_gemvx_int2(c->layout, c->mode, c->mb, rd)
                         : pick_gemvx_int1(c->layout, c->mode, c->mb, rd);
  if (!c->fn) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemvx: no member for bits=%d layout=%d mode=%d mb=%d rd=%d", c->bits, c->layout, c->mode, c->mb, rd);
    return WQAA_ERR_UNSUPPORTED;
  }
  return WQAA_OK;
}

int gemvx_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan) {
  GemvxChoice c;
  int st = gemvx_choose(d, m, &c);
  if (st != WQAA_OK) return st;
  if (plan) {
    plan->kernel_family = 1;
    plan->block_m = c.mb;
    plan->block_n = c.R * (c.nw / c.kw);   // rows a workgroup works on at a time
    plan->block_k = 64 * c.E * c.D;
    plan->threads = c.nw * 64;
    plan->grid = c.grid;
    plan->rows_per_wave = c.R;
    plan->batch_tile = c.mb;
    plan->pipeline_depth = c.D;
    plan->split_k = c.kw;
    plan->lds_bytes = c.lds;
    char wd[24];
    short_wdtype(d, wd, sizeof(wd));
    snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_gemvx_b%dr%dd%dk%d", m, d.N, d.K, short_dtype(d.a_dtype), wd,
             c.mb, c.R, c.D, c.kw);
    if (c.areg) strncat(plan->name, "_areg", sizeof(plan->name) - strlen(plan->name) - 1);
  }
  return WQAA_OK;
}

static void gemvx_fill(const wqaa_matmul_desc& d, const GemvxChoice& c, const void* A, const void* B, const void* Scale,
                       const void* Zeros, const void* Bias, void* C, int m, GemvxArgs* out) {
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  GemvxArgs& a = *out;
  a.A = A; a.B = B; a.scale = Scale; a.zeros = Zeros; a.bias = Bias; a.C = C;
  a.m = m; a.N = d.N; a.K = d.K;
  a.kg = d.K / g;
  {
    const int dq = g / c.E > 0 ? g / c.E : 1;
    a.gq_shift = ilog2_exact(dq);
    a.gq_magic = a.gq_shift >= 0 ? 0u : (uint32_t)(((1ull << 32) + dq - 1) / dq);
  }
  a.nc = c.nc; a.cpr = c.cpr; a.nsteps = c.nsteps; a.kw = c.kw;
  a.row_bytes = (long)d.K * c.bits / 8;
  a.has_bias = d.with_bias;
  a.out_dtype = d.out_dtype;
  const bool is_signed = d.w_format == WQAA_W_INT;
  a.zint = is_signed ? (c.bits == 1 ? 1 : (1 << (c.bits - 1))) : 0;
  a.flip = (is_signed && c.bits == 1) ? 0xFFFFFFFFu : 0u;
  a.zq_row_bytes = d.N * c.bits / 8;
  const int slots = c.nw / c.kw;
  a.n_rgb = ((d.N + c.R - 1) / c.R + slots - 1) / slots;
  a.slots = slots;
  a.kw_magic = (65536u + (uint32_t)c.kw - 1u) / (uint32_t)c.kw;
  a.residual = nullptr;
  a.norm_weight = nullptr;
  a.norm_eps = 0.f;
  a.norm_inv_k = 1.f / (float)d.K;
}

static void gemvx_set_norm(GemvxArgs& a, const wqaa_epilogue* epi) {
  a.norm_weight = epi->norm_weight;
  a.norm_eps = epi->norm_eps;
}

static int gemvx_dispatch(const GemvxChoice& c, GemvxGroupArgs& ga, int grid_x, int count, hipStream_t stream, hipEvent_t start,
                          hipEvent_t stop) {
  void* params[] = {&ga};
  dim3 grid(grid_x, count, 1), block(c.nw * 64, 1, 1);
  hipError_t e;
  if (start || stop) e = hipExtLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream, start, stop, 0);
  else e = hipLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream);
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "gemvx launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

int gemvx_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* Scale, const void* Zeros,
                 const void* Bias, void* C, int m, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi) {
  // WQAA_EPI_ADD_RESIDUAL or WQAA_EPI_RMSNORM_INPUT (one of them: checked by the caller)
  const int pro = epi == nullptr ? 0 : (epi->flags & WQAA_EPI_RMSNORM_INPUT) ? 3 : 1;
  GemvxChoice c;
  {
    static thread_local ChoiceMemo<GemvxChoice> memo;
    const int q = pro == 3 ? 10 : pro ? 8 : 7;
    if (const GemvxChoice* hit = memo.find(d, m, q)) {
      c = *hit;
    } else {
      int st = gemvx_choose(d, m, &c, pro);
      if (st != WQAA_OK) return st;
      memo.put(d, m, q, c);
    }
  }
  GemvxGroupArgs ga;
  gemvx_fill(d, c, A, B, Scale, Zeros, Bias, C, m, &ga.p[0]);
  if (pro == 1) ga.p[0].residual = epi->residual;
  if (pro == 3) gemvx_set_norm(ga.p[0], epi);
  return gemvx_dispatch(c, ga, c.grid, 1, stream, start, stop);
}

// ---- gate_proj + up_proj + the gated activation in one launch (wqaa_matmul_gate_up) ---------------------------------------
// Tile configuration: what the selector gives the two projections concatenated (2 N rows, two rows per wave); a wave's
// two rows are row n of each.  args.N stays the projections' N: a row group IS an output element.
static int gemvx_pair_choose(const wqaa_matmul_desc& d, int m, GemvxChoice* c, bool norm) {
  if (!gemvx_covers(d, m) || d.out_dtype != WQAA_F16) {
    set_error(WQAA_ERR_UNSUPPORTED, "matmul_gate_up: needs float16 activations, 1/2/4-bit integer weights, float16 output and m <= 2 (got m=%d)", m);
    return WQAA_ERR_UNSUPPORTED;
  }
  wqaa_matmul_desc merged = d;
  merged.N = 2 * d.N;
  static thread_local ChoiceMemo<GemvxChoice> memo;
  const int q = norm ? 11 : 9;
  if (const GemvxChoice* hit = memo.find(merged, m, q)) {
    *c = *hit;
    return WQAA_OK;
  }
  // the K split across waves is the fp32 summation order of a row: the one each projection gets ALONE, so that the pair's g and
  // u are the float16 values the projections' own launches store (rows per wave, workgroup width and grid do not change a row's bits)
  GemvxChoice alone;
  int st = gemvx_choose(d, m, &alone, 1);
  if (st != WQAA_OK) return st;
  st = gemvx_choose(merged, m, c, norm ? 4 : 2, alone.kw);
  if (st == WQAA_OK) memo.put(merged, m, q, *c);
  return st;
}

int gemvx_pair_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan, bool norm) {
  GemvxChoice c;
  int st = gemvx_pair_choose(d, m, &c, norm);
  if (st != WQAA_OK || !plan) return st;
  wqaa_matmul_desc merged = d;
  merged.N = 2 * d.N;
  memset(plan, 0, sizeof(*plan));
  plan->kernel_family = 1;
  plan->block_m = c.mb;
  plan->block_n = c.nw / c.kw;             // output elements a workgroup works on at a time
  plan->block_k = 64 * c.E * c.D;
  plan->threads = c.nw * 64;
  plan->grid = c.grid;
  plan->rows_per_wave = c.R;
  plan->batch_tile = c.mb;
  plan->pipeline_depth = c.D;
  plan->split_k = c.kw;
  plan->lds_bytes = c.lds;
  char wd[24];
  short_wdtype(d, wd, sizeof(wd));
  snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_gemvx_b%dr%dd%dk%d_pair%s", m, d.N, d.K, short_dtype(d.a_dtype), wd, c.mb,
           c.R, c.D, c.kw, norm ? "_norm" : "");
  return WQAA_OK;
}

int gemvx_pair_launch(const wqaa_matmul_desc& d, const wqaa_group_item* gate, const wqaa_group_item* up, void* act, int m,
                      hipStream_t stream, const wqaa_epilogue* norm) {
  GemvxChoice c;
  int st = gemvx_pair_choose(d, m, &c, norm != nullptr);
  if (st != WQAA_OK) return st;
  GemvxGroupArgs ga;
  // filled as the 2 N-row operator (row-group blocks, K split), then N put back: rows of a pair are indexed by output element
  wqaa_matmul_desc merged = d;
  merged.N = 2 * d.N;
  gemvx_fill(merged, c, gate->A, gate->B, gate->Scale, gate->Zeros, gate->Bias, act, m, &ga.p[0]);
  gemvx_fill(merged, c, up->A, up->B, up->Scale, up->Zeros, up->Bias, act, m, &ga.p[1]);
  for (int i = 0; i < 2; ++i) {
    ga.p[i].N = d.N;
    ga.p[i].zq_row_bytes = d.N * c.bits / 8;
    if (norm) gemvx_set_norm(ga.p[i], norm);
  }
  return gemvx_dispatch(c, ga, c.grid, 1, stream, nullptr, nullptr);
}

// ---- a group of independent operators in one launch (wqaa_matmul_group) -------------------------------------------------
// The tile configuration is the one the selector gives the MERGED operator (N = sum of the members' rows: what a caller
// that concatenates q/k/v or gate/up into one Linear would get); every member then takes gridDim.x workgroups of it.
// The K split across waves member i gets ALONE - the fp32 summation order of its rows in a single call.  Round 6: a group keeps it per
// member (GemvxArgs::kw / slots / n_rgb are per member; the workgroup width is the group's), so members of very different widths -
// q (8192 rows) next to k / v (1024 rows each, K split in two when alone) under grouped-query attention - fuse and still give the
// bits of their single calls.  (Until round 5 every member had to land on the merged operator's split: a 70B layer's q/k/v ran as
// three launches, 16.1 us where the fused launch takes ~11.5.)
static int solo_kw(const wqaa_matmul_desc& merged, int N, int m, int* D = nullptr) {
  wqaa_matmul_desc d = merged;
  d.N = N;
  static thread_local ChoiceMemo<GemvxChoice> memo;
  GemvxChoice c;
  if (const GemvxChoice* hit = memo.find(d, m, 15)) {
    c = *hit;
  } else {
    if (gemvx_choose(d, m, &c, 0) != WQAA_OK) return 0;
    memo.put(d, m, 15, c);
  }
  if (D) *D = c.D;
  return c.kw;
}

static int gemvx_group_choose(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, GemvxChoice* c, int* grid_x, bool norm = false,
                              int* kws = nullptr) {
  {
    static thread_local ChoiceMemo<GemvxChoice> memo;
    const int q = (norm ? 48 : 16) + count;
    if (const GemvxChoice* hit = memo.find(merged, m, q)) {
      *c = *hit;
    } else {
      int st = gemvx_choose(merged, m, c, norm ? 3 : 0);
      if (st != WQAA_OK) return st;
      memo.put(merged, m, q, *c);
    }
  }
  int need = 1;
  for (int i = 0; i < count; ++i) {
    int kw = solo_kw(merged, Ns[i], m);
    if (kw < 1 || c->nw % kw != 0) kw = c->kw;        // (gemvx_group_eligible has refused such groups: defensive)
    if (kws) kws[i] = kw;
    const int slots = c->nw / kw;
    const int n_rgb = ((Ns[i] + c->R - 1) / c->R + slots - 1) / slots;
    if (n_rgb > need) need = n_rgb;
  }
  // every member gets the workgroups of its LARGEST member (one row-group block each; the shorter members' surplus
  // workgroups leave at once).  No cap at what the chip holds at once: same-call A/B of the gate/up group of the
  // headline step (2 x 688 blocks): 1024 workgroups taking up to two blocks each 11.94 us, 1376 workgroups 10.81 us -
  // the hardware dispatcher balances better than a static second round (tools/ab_group.py)
  int gx = need;
  { int v; if (gemv_knob("group_grid", &v) && v > 0 && v < gx) gx = v; }    // tuning aid: workgroups per member
  if (gx >= 8) gx = (gx + 7) / 8 * 8;                 // whole XCD rounds per member: blockIdx.x % 8 stays the XCD
  *grid_x = gx;
  return WQAA_OK;
}

// A fused group must give every member the bits a single call would: the family (exact products vs per-element rounding) is
// chosen from N, so each member ALONE has to be this family's; its K split across waves (the fp32 summation order of a row) is kept
// per member and must divide the group's workgroup width.  (Rows per wave, workgroup width, grid and register-resident vs
// LDS-staged activations do not change a row's arithmetic: tests/test_group_gpu.py.)
bool gemvx_group_eligible(const wqaa_matmul_desc& merged, const wqaa_matmul_desc* const* descs, int count, int m, bool norm) {
  // (the norm in front exists in this family only: what it covers counts, not where it is the faster one)
  auto takes = [&](const wqaa_matmul_desc& d) { return norm ? gemvx_covers(d, m) : gemvx_eligible(d, m); };
  if (count < 1 || count > kGemvxGroupMax || !takes(merged)) return false;
  GemvxChoice cm;
  if (gemvx_choose(merged, m, &cm, norm ? 3 : 0) != WQAA_OK) return false;
  for (int i = 0; descs && i < count; ++i) {
    // (the norm's capacity - the rows within the items a workgroup loads ahead - is the MERGED configuration's: a narrow member,
    // the k / v of grouped-query attention, runs in the group's workgroups)
    if (!takes(*descs[i])) return false;
    int Di = 0;
    const int kwi = solo_kw(merged, descs[i]->N, m, &Di);
    if (kwi < 1 || Di != cm.D || cm.nw % kwi != 0) return false;
  }
  return true;
}

int gemvx_group_plan(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, wqaa_plan* plan) {
  GemvxChoice c;
  int gx = 0;
  int st = gemvx_group_choose(merged, Ns, count, m, &c, &gx);
  if (st != WQAA_OK) return st;
  st = gemvx_plan(merged, m, plan);
  if (st == WQAA_OK && plan) {
    plan->grid = gx * count;
    char tail[16];
    snprintf(tail, sizeof(tail), "_x%d", count);
    strncat(plan->name, tail, sizeof(plan->name) - strlen(plan->name) - 1);
  }
  return st;
}

int gemvx_group_launch(const wqaa_matmul_desc& merged, const wqaa_group_item* items, int count, int m, hipStream_t stream,
                       const wqaa_epilogue* norm) {
  int Ns[kGemvxGroupMax], kws[kGemvxGroupMax];
  for (int i = 0; i < count; ++i) Ns[i] = items[i].desc->N;
  GemvxChoice c;
  int gx = 0;
  int st = gemvx_group_choose(merged, Ns, count, m, &c, &gx, norm != nullptr, kws);
  if (st != WQAA_OK) return st;
  GemvxGroupArgs ga;
  for (int i = 0; i < count; ++i) {
    GemvxChoice ci = c;
    ci.kw = kws[i];                                    // the member's own K split: its single call's summation order
    gemvx_fill(*items[i].desc, ci, items[i].A, items[i].B, items[i].Scale, items[i].Zeros, items[i].Bias, items[i].C, m, &ga.p[i]);
    if (norm) gemvx_set_norm(ga.p[i], norm);          // one norm for the group: its members read the same hidden state
  }
  return gemvx_dispatch(c, ga, gx, count, stream, nullptr, nullptr);
}

void gemvx_init() {
  for (int bits : {4, 2, 1})
    for (int layout = 0; layout < 2; ++layout)
      for (int mode = 0; mode <= MD_ZQ; ++mode)
        for (int mb = 1; mb <= 2; ++mb)
          for (int rd : {12, 22, 13, 23}) {
            gemvx_fn fn = bits == 4 ? pick_gemvx_int4(layout, mode, mb, rd) : bits == 2 ? pick_gemvx_int2(layout, mode, mb, rd)
                                                                                       : pick_gemvx_int1(layout, mode, mb, rd);
            if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            for (int prd : {rd, 1000 + rd}) {
              fn = bits == 4 ? pick_gemvx_pro4(layout, mode, mb, prd) : bits == 2 ? pick_gemvx_pro2(layout, mode, mb, prd)
                                                                                : pick_gemvx_pro1(layout, mode, mb, prd);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            for (int prd : {3000 + rd, 4000 + rd}) {
              fn = bits == 4 ? pick_gemvx_norm4(layout, mode, mb, prd) : bits == 2 ? pick_gemvx_norm2(layout, mode, mb, prd)
                                                                                 : pick_gemvx_norm1(layout, mode, mb, prd);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
          }
  (void)hipGetLastError();
}

}  // namespace wqaa
