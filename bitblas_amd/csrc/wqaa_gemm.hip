// wqaa_gemm.hip - host side of the MFMA GEMM family: tile-config selector, split-K scratch, launch.
// The kernels live in wqaa_gemm_kernel.h; the member tables are instantiated in wqaa_gemm_inst_*.hip.
#include "wqaa_gemm_mid_kernel.h"

#include <vector>

namespace wqaa {

// Scratch for the fp32 / int32 partial sums of the split-K members.  Ownership (the reference leaves workspace
// ownership with the caller: bitblas/ops/general_matmul/__init__.py:29, 456-457, 482):
//   * caller-owned: wqaa_matmul_opts(..., workspace, workspace_bytes) - the library allocates nothing;
//   * otherwise a library pool with ONE slab per (device, stream): two streams never share partial sums, and a slab
//     that has to grow is RETIRED, never freed - a hipGraph captured earlier keeps replaying into memory that is
//     still allocated.  Growing calls hipMalloc, which is illegal during stream capture: that case is refused with
//     an error telling the caller to run the shape once outside capture (or to pass its own workspace).
struct WsSlab {
  int dev;
  hipStream_t stream;
  void* ptr;
  size_t bytes;
};
static std::vector<WsSlab> g_ws;
static std::vector<void*> g_ws_retired;
static std::mutex g_ws_mu;
void* pool_workspace(hipStream_t stream, size_t bytes) {
  const int dev = current_device();
  if (dev < 0) return nullptr;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  WsSlab* slab = nullptr;
  for (auto& w : g_ws)
    if (w.dev == dev && w.stream == stream) { slab = &w; break; }
  if (slab && slab->bytes >= bytes) return slab->ptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) {
    set_error(WQAA_ERR_LAUNCH, "gemm: the scratch of this stream (split-K partial sums / B_decode) has to grow to %zu B, which cannot happen during stream "
              "capture: run this shape once outside capture, or pass a workspace (wqaa_matmul_opts)", bytes);
    return nullptr;
  }
  size_t want = bytes < (8u << 20) ? (8u << 20) : bytes;
  if (slab && want < 2 * slab->bytes) want = 2 * slab->bytes;     // geometric growth bounds the retired total
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    set_error(WQAA_ERR_LAUNCH, "gemm: cannot allocate %zu B of split-K scratch", want);
    return nullptr;
  }
  if (slab) {
    g_ws_retired.push_back(slab->ptr);
    slab->ptr = p;
    slab->bytes = want;
  } else {
    g_ws.push_back(WsSlab{dev, stream, p, want});
  }
  return p;
}

// true where pool_workspace(stream, bytes) would succeed: the slab is there, or the stream is not capturing (it can grow)
bool pool_workspace_ready(hipStream_t stream, size_t bytes) {
  const int dev = current_device();
  if (dev < 0) return false;
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto& w : g_ws)
      if (w.dev == dev && w.stream == stream && w.bytes >= bytes) return true;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  return cs == hipStreamCaptureStatusNone;
}

static gemm_fn pick_gemm(int kind, int layout, int at, int mode, int flags, int mf) {
  if (at == AT_F16 && (flags & FL_BF16)) return layout == LAYOUT_PLAIN ? pick_gemm_bf16(kind, mode, mf) : nullptr;
  if (at == AT_F16) {
    switch (kind) {
      case DK_INT4: return pick_gemm_f16_int4(layout, mode, mf);
      case DK_INT2:
      case DK_INT1: return pick_gemm_f16_int21(kind, layout, mode, mf);
      default: return pick_gemm_f16_other(kind, mode, flags, mf);
    }
  }
  if (mode != MD_NONE) return nullptr;
  return pick_gemm_i8_f8(kind, layout, at, flags, mf);
}

struct GemmChoice {
  gemm_fn fn;
  int kind, layout, at, mode, flags, bits;
  int mf, ks, kl, nwaves, bn, ksplit, skinny, decode, wide, pp, pp_shift, pp_bm;
  int tiles_m, tiles_n, lds;
  int fp4_table;
  // the remainder of a partial round of 256 x 256 tiles as a second launch of the 128 x 256 member over the last N-tiles
  gemm_fn tail_fn;
  int tail_lds, tail_tiles_m, tail_tiles_n;
  int decode_long;      // the one-launch decode member stages the wave's whole k-range (M-sized slots) and walks units (fragment, k-block)
  int decode_grid;      // > 0: the persistent form of the one-launch decode member (grid < number of 16-row fragments)
  int decode_kslice;    // the K-sliced form of the decode member: 8 slices x decode_grid / 8 workgroups, fp32 partial sums + wq_mid_reduce_kernel
  int pp_avail;         // m > 128: a fused ping-pong member takes this descriptor (whether or not the round estimate chose it here)
  int mid;              // the mid-M one-launch split-K member (wqaa_gemm_mid_kernel.h); mid_nkh: k-steps per k-half of a slice
  int mid_nkh;
};

static int gemm_choose(const wqaa_matmul_desc& d, int m, GemmChoice* c, bool fused_epilogue = false, bool no_mid = false) {
  const int a = d.a_dtype;
  c->mid = 0;
  c->mid_nkh = 0;
  c->fp4_table = 0;
  c->tail_fn = nullptr;
  c->tail_lds = c->tail_tiles_m = c->tail_tiles_n = 0;
  c->pp_avail = 0;
  c->decode_grid = 0;
  c->decode_long = 0;
  c->decode_kslice = 0;
  c->flags = 0;
  c->layout = d.w_layout == WQAA_LAYOUT_LOP3 ? LAYOUT_LOP3 : LAYOUT_PLAIN;
  if (a == WQAA_F16) c->at = AT_F16;
  else if (a == WQAA_BF16) { c->at = AT_F16; c->flags |= FL_BF16; }   // same machine path, bfloat16 arithmetic
  else if (a == WQAA_I8) c->at = AT_I8;
  else if (a == WQAA_I4) c->at = AT_I4;
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
      c->bits = (a == WQAA_F16 || a == WQAA_BF16) ? 16 : 8;
      if (a == WQAA_I4) { c->kind = DK_INT4; c->bits = 4; }   // two's-complement nibbles
      break;
    default: c->kind = -1;
  }
  if (a == WQAA_I4 && !(c->kind == DK_INT4 && d.w_format == WQAA_W_NATIVE) && c->kind != DK_INT2) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: int4 activations pair with int4 (native) or 2-bit weights only");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->kind < 0) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: weight format %d / %d bits not supported", d.w_format, d.w_bits);
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->at == AT_F8 && c->kind != DK_E4M3 && c->kind != DK_E5M2) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: fp8 activations need fp8 weights");
    return WQAA_ERR_UNSUPPORTED;
  }
  if (c->kind == DK_E4M3 && d.strict_reference && c->at == AT_F16 && !(c->flags & FL_BF16)) c->flags |= FL_STRICT;
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
  const bool k_ok = d.K % c->ks == 0;      // (the ping-pong members have their own K grid: checked after them)
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
  const int nsteps = d.K / c->ks;
  // The ping-pong members (wqaa_gemm_pp_kernel.h), where one exists: 4-bit weights x float16, 2-bit weights x int8, dense fp8;
  // four k-tiles (256 / 512 / 128 k) per trip, groups of 128 * 2^i (Scale / Zeros rows 4-byte aligned: K / g even), float16 /
  // int32 output through LDS (N a multiple of 8), 32-bit buffer offsets.  WQAA_GEMM_TUNE=pp_tile=0: the lockstep members only.
  // Which tile: a workgroup is alone on its CU, so time goes by ROUNDS of the chip.  Same-box medians, uint4 g128 + zeros, K =
  // 4096 (profiles/r03_lab_pp128.txt): a round of x 256 x 256 tiles 80 + 0.113 x us (109 full: the chip is power-limited), of
  // 128 x 256 tiles 55 + 0.052 x (68.5 full), the lockstep 128 x 128 member 12 + 40 per round of 256 tiles, in half rounds.  All
  // three scale with K and, near enough, together with the formats - only the ratios matter here.
  c->pp = 0;
  c->pp_shift = 0;
  c->pp_bm = 0;
  if (m > 128 && d.N >= 128) {
    const bool dense16 = c->kind == DK_NATIVE && c->at == AT_F16;        // float16 / bfloat16 x the same type: the dense fp8 skeleton on 16-bit lines
    const bool dense8 = c->kind == DK_NATIVE && c->at == AT_I8;          // int8 x int8: the same skeleton, 128 k per line
    const int kb = dense16 ? 64 : c->at == AT_F16 ? 256 : (c->at == AT_F8 || dense8) ? 128 : 512;   // k per trip of the main loop
    const int gb = c->at == AT_F8 ? 1 : g / (kb / 2);                   // k-bodies (two k-tiles) per group
    // (packed integer zero points - GPTQ checkpoints: a wave fetches the 16 bytes of its 32 rows per group, so N in whole waves)
    // (one group per row - per-channel scales, the reference's default group_size = -1 - is the one odd K / g the members take:
    // a row's 16-byte window then opens on the even element in front of it)
    const bool one_group = d.K / g == 1;
    const bool meta_ok = c->mode == MD_NONE || (g % (kb / 2) == 0 && (one_group || (ilog2_exact(gb) >= 0 && ((d.K / g) & 1) == 0)) &&
                                                (long)d.N * (d.K / g) >= 8 && (c->mode != MD_ZQ || (d.N % 32 == 0 && c->bits == 4)));
    // (the caller's fused epilogue - int32 sums / row scale / tensor scale -> float16 - rides in the integer members' output stage)
    const bool epi_ok = !fused_epilogue || (c->at == AT_I8 && d.out_dtype == WQAA_F16 && d.a_dtype == WQAA_I8 && c->kind != DK_NATIVE);
    const bool out_ok = c->at == AT_I8 ? (fused_epilogue ? d.out_dtype == WQAA_F16 : d.out_dtype == WQAA_I32)
                        : c->at == AT_F8 ? d.out_dtype == WQAA_F16
                        : dense16 ? (!d.with_bias && d.out_dtype == ((c->flags & FL_BF16) ? WQAA_BF16 : WQAA_F16))      // (its output pass: 2-byte elements, no bias)
                        : (d.out_dtype == WQAA_F32 || d.out_dtype == ((c->flags & FL_BF16) ? WQAA_BF16 : WQAA_F16));
    const long a_bytes = (long)m * d.K * (c->at == AT_F16 ? 2 : 1), w_bytes = (long)d.N * d.K * c->bits / 8;
    const bool shape_ok = epi_ok && d.K % kb == 0 && meta_ok && out_ok && d.N % 8 == 0 && a_bytes + 256L * d.K * 2 < (1L << 31) &&
                          w_bytes < (1L << 31) && d.k_split_hint <= 1 && !gemm_knob_set("ksplit");
    auto rounds_time = [&](long tiles, double base, double slope) {
      const long full = tiles / cus_, rem = tiles % cus_;
      return (double)full * (base + slope * cus_) + (rem ? base + slope * (double)rem : 0.0);
    };
    const long tiles_n256 = (d.N + 255) / 256;
    const double t256 = rounds_time((long)((m + 255) / 256) * tiles_n256, 80.0, 0.113);
    const double t128 = rounds_time((long)((m + 127) / 128) * tiles_n256, 55.0, 0.052);
    const long tiles_l = (long)((m + 127) / 128) * ((d.N + 127) / 128);
    // (the lockstep int8 / fp8 members are further behind their ping-pong counterparts than the fp16 one: tools/ab_pp_tile.py,
    // profiles/r03_ab_pp_tile_*.txt - int2 x int8 M = 1024 4096^2 39.6 vs 34.4 us on the 128-row tile, e4m3 1024 x 8192 x 8192 106 vs 84)
    // (and the dense 16-bit ones: float16 1024 x 4096^2 75.5 us on the lockstep member, 49.4 on the 128-row tile; profiles/r04_ab_pp_tile_dense.txt)
    const double tlock = (12.0 + 40.0 * 0.5 * (double)((2 * tiles_l + cus_ - 1) / cus_)) * (c->at == AT_I8 ? 1.35 : c->at == AT_F8 ? 1.3 : dense16 ? 1.4 : 1.0);
    int lds256 = 0, lds128 = 0, ldss = 0;
    gemm_fn fn256 = shape_ok && m >= 256 && d.N >= 256 ? pick_gemm_pp(c->kind, c->layout, c->at, c->mode, c->flags, 256, 256, &lds256) : nullptr;
    gemm_fn fn128 = shape_ok && d.N >= 256 ? pick_gemm_pp(c->kind, c->layout, c->at, c->mode, c->flags, 128, 256, &lds128) : nullptr;
    // dense fp8 also has a 128 x 128 tile, for outputs that give the CUs no wider one each (4096 x 1024: the c5 column shards)
    gemm_fn fns = shape_ok ? pick_gemm_pp(c->kind, c->layout, c->at, c->mode, c->flags, 128, 128, &ldss) : nullptr;
    c->pp_avail = (fn256 || fn128 || fns) ? 1 : 0;
    // (a round of x 128 x 128 fp8 tiles: 30 + 0.035 x in the units of the estimates above; profiles/r03_ab_pp_tile_f8_128.txt)
    const double ts = rounds_time(tiles_l, 30.0, 0.035);
    // A shape that leaves the last round of 256 x 256 tiles mostly empty (2048 x 11008: 344 tiles = one round + 88) pays a whole
    // tile's latency for it (80 + 0.113 x).  The columns of that remainder go out as a SECOND launch of the 128-row tile - twice
    // the workgroups, 55 + 0.052 x - behind a launch of whole rounds: N-tiles [0, n_main) by the first, the rest by the second
    // (GemmArgs::tile_n_off); + ~3 us of boundary.  WQAA_GEMM_TUNE=pp_tail=0: never.
    int hy_n_main = 0;
    double thy = 1e30;
    {
      int tail_on = 1;
      (void)gemm_knob("pp_tail", &tail_on);
      const long tm256 = (m + 255) / 256, tm128 = (m + 127) / 128;
      const long T = tm256 * tiles_n256;
      if (fn256 && fn128 && tail_on != 0 && T > cus_ && T % cus_ != 0 && tiles_n256 >= 2) {
        const long n_fit = ((T / cus_) * cus_) / tm256;
        for (long nm = n_fit; nm >= n_fit - 1 && nm >= 1; --nm) {
          if (nm >= tiles_n256) continue;
          const double t = rounds_time(tm256 * nm, 80.0, 0.113) + rounds_time(tm128 * (tiles_n256 - nm), 55.0, 0.052) + 3.0;
          if (t < (nm == n_fit ? thy : 0.99 * thy)) { thy = t; hy_n_main = (int)nm; }      // (ties: the whole rounds to the first launch)
        }
      }
    }
    int bm = 0, bn = 256;
    int forced_bm = 0, forced_bn = 0;
    if (gemm_knob("pp_tile", &forced_bm)) {                        // tuning aid: force a tile (pp_tile=0: the lockstep members; 128 with
      bm = forced_bm;                                              // pp_bn=128: the 128 x 128 tile)
      (void)gemm_knob("pp_bn", &forced_bn);
      if (bm == 128 && forced_bn == 128 && fns) bn = 128;
      else if ((bm == 256 && !fn256) || (bm == 128 && !fn128)) bm = 0;
    } else {
      double best = k_ok ? 0.97 * tlock : 1e30;      // (K off the lockstep members' grid: any ping-pong member that takes it)
      if (fn256 && t256 < best) { bm = 256; best = t256; }
      if (fn128 && t128 < best) { bm = 128; best = t128; }
      if (fns && ts < best) { bm = 128; bn = 128; best = ts; }
      if (hy_n_main > 0 && thy < 0.97 * best) { bm = 256; bn = 256; best = thy; }
      else hy_n_main = 0;
    }
    if (bm != 256 || bn != 256) hy_n_main = 0;       // (a forced tile stays what it says)
    if (bm) {
      c->pp = 1;
      c->pp_bm = bm;
      c->pp_shift = c->mode == MD_NONE ? 0 : one_group ? 20 : ilog2_exact(gb);     // (one group: every k-body maps to group 0)
      c->fn = bn == 128 ? fns : bm == 256 ? fn256 : fn128;
      c->mf = bm / 16;
      c->nwaves = 8;
      c->bn = bn;
      c->skinny = 0;
      c->decode = 0;
      c->wide = 0;
      c->ks = c->at == AT_F16 ? 64 : 128;
      c->tiles_m = (m + bm - 1) / bm;
      c->tiles_n = (d.N + bn - 1) / bn;
      c->lds = bn == 128 ? ldss : bm == 256 ? lds256 : lds128;
      c->ksplit = 1;
      if (hy_n_main > 0) {
        c->tail_fn = fn128;
        c->tail_lds = lds128;
        c->tail_tiles_m = (m + 127) / 128;
        c->tail_tiles_n = c->tiles_n - hy_n_main;
        c->tiles_n = hy_n_main;
      }
      return WQAA_OK;
    }
  }
  if (!k_ok) {
    set_error(WQAA_ERR_UNSUPPORTED, "gemm: K=%d must be a multiple of %d", d.K, c->ks);
    return WQAA_ERR_UNSUPPORTED;
  }
  // round 5 - the mid-M member (wqaa_gemm_mid_kernel.h): 4-bit weights x float16, up to 128 rows per M-tile, K = 2048 nkh with the
  // workgroup's whole activation slice (16 mf rows x K / 8) in LDS; K in 8 slices, everything a workgroup reads asked for at once.
  // WHERE, from same-process A/B against the members it stands in for (tools/r05_ab_mid.py, profiles/r05_ab_mid_v3.txt; uint4 g128 +
  // zeros, us): 65 ... 128 rows in one round of the chip at K = 4096: M = 96 16.1 vs 16.7, M = 128 16.9 vs 17.1; up to 64 rows where
  // the old members ran long k ranges or several rounds: 64 x 4096 x 8192 17.9 vs 21.1, 32 x 4096 x 8192 13.8 vs 15.2, 64 x 8192 x 4096
  // 18.4 vs 20.9, 64 x 11008 x 4096 23.9 vs 26.4.  It LOSES at up to 64 rows in one round at K = 4096 (M = 32 11.0 vs 9.6, M = 64 13.8 vs
  // 12.3: the split-K skinny member's smaller workgroups overlap), at 128 rows over several rounds (11008 x 4096 39.0 vs 33.8) and at
  // K = 2048 (16.4 vs 13.1) - those keep their members.  WQAA_GEMM_TUNE=mid=0: never; mid=2: wherever the shape fits (the
  // parity tests run every instantiation that way).
  if (!no_mid && !fused_epilogue && c->at == AT_F16 && !(c->flags & FL_BF16) && c->kind == DK_INT4 && d.k_split_hint <= 1 &&
      !gemm_knob_set("ksplit") && m > 16) {
    int mid_knob = 1;                                         // WQAA_GEMM_TUNE=mid=0: never; mid=2: wherever a member exists (tests)
    (void)gemm_knob("mid", &mid_knob);
    const bool force = mid_knob == 2;
    const int nkh = d.K % 2048 == 0 ? d.K / 2048 : 0;
    const int tm = (m + 127) / 128;
    const int rows = (m + tm - 1) / tm;                       // rows per M-tile
    const int mf = rows > 64 ? 8 : rows > 32 ? 4 : 2;
    const long tiles = (long)((m + 16 * mf - 1) / (16 * mf)) * ((d.N + 127) / 128);
    const long wgs = tiles * 8;
    int lds = 0;
    gemm_fn fn = (nkh == 2 || nkh == 4) && mf * nkh <= 16 ? pick_gemm_mid(c->kind, c->layout, c->mode, mf, nkh, &lds) : nullptr;
    // (the members with NKH >= 2 fetch Scale / Zeros of their NKH consecutive groups in one load per row: one group per k-step,
    // rows aligned; other group sizes keep the members they had)
    const bool widemeta = (c->mode == MD_S || c->mode == MD_ZO || c->mode == MD_ZR) && nkh >= 2;
    const bool meta_ok = !widemeta || (g == c->ks && (d.K / g) % nkh == 0);
    // (128 rows on half a chip of workgroups - N = 2048: 16.0 vs 14.1 us - keeps its member: at least three quarters of a round)
    const bool measured = m <= 128 && (mf == 8 ? (nkh == 2 && wgs <= cus_ && 4 * wgs >= 3L * cus_)
                                               : (nkh == 4 ? wgs <= 3L * cus_ : (nkh == 2 && wgs > cus_ && wgs <= 3L * cus_)));
    if (fn && meta_ok && mid_knob != 0 && (force || measured) && tiles <= 256 &&
        (long)m * d.K * 2 < (1L << 32) && d.K < (1 << 23) && m < (1 << 23) && (c->mode != MD_ZQ || d.N % 2 == 0)) {
      c->mid = 1;
      c->mid_nkh = nkh;
      c->fn = fn;
      c->mf = mf;
      c->nwaves = 8;
      c->bn = 128;
      c->skinny = c->decode = c->wide = 0;
      c->tiles_m = (m + 16 * mf - 1) / (16 * mf);
      c->tiles_n = (d.N + 127) / 128;
      c->lds = lds;
      c->ksplit = 1;              // (no split-K reduce of the old kind: the member brings its own second launch, or none)
      return WQAA_OK;
    }
  }
  // small decode batches: one launch with K split across the 8 waves of a workgroup, no partial sums
  // in memory (WQAA_GEMM_TUNE=decode=0: back to the split-K skinny member + reduce launch).  Every
  // workgroup reads all M activation rows, so it pays only while M and the number of 16-row weight
  // fragments are small.  Same-box A/B against the skinny member, uint4 g128 + zeros: 4096^2 M=5
  // 7.6 vs 9.2 us, M=8 8.2 vs 9.3, M=16 9.5 vs 9.9, M=32 13.2 vs 11.6; 11008x4096 M=8 18.3 vs 16.2;
  // 4096x11008 M=8 15.2 vs 17.8; int2 x int8 4096^2 M=5 5.7 vs 7.3.
  c->decode = 0;
  // With the activations through LDS-DMA (member 211) the one-launch member wins whenever the 16-row weight fragments
  // fill whole rounds of the chip - one 136 KiB workgroup per CU - and loses the tail of a partial round to the
  // skinny member, whose small workgroups overlap.  Same-call A/B, uint4 g128 + zeros, member vs skinny + reduce
  // (profiles/r02_ab_decode_fence.txt): M=16 N=2048 6.0 vs 7.4 us, N=3072 6.3 vs 8.7, N=4096 6.4 vs 9.0, N=5120 (1.25
  // rounds) 11.0 vs 10.4, N=8192 (2 rounds) 11.6 vs 12.6, N=11008 (2.7 rounds) 16.7 vs 15.1; 4096x11008 16.5 vs 19.7;
  // M=8 N=6144 (1.5 rounds) 9.8 vs 10.2, N=8192 10.4 vs 11.3, N=11008 14.5 vs 13.5, 4096x11008 12.1 vs 18.8;
  // M=4 8192^2 15.9 vs 17.3.
  const int frags = (d.N + 15) / 16;
  const int decode_max_m = 16;
  // (the direct-load member, which only packed-int4 activations still use: M <= 8 up to 1.5 rounds, M = 9..16 between
  // 0.75 and 1 round - the round-1 table above)
  // (two whole rounds of fragments at M = 9 ... 16: a tie with the split-K skinny member at K = 8192 - 19.2 vs 19.4 us at 8192^2 - and
  // behind it on longer K, where the activations' L2 traffic of the second round costs more than the reduce launch: 8192 x 28672
  // M = 16 59.4 vs 54.2 us, profiles/r04_decode_longk.txt; measured for 16-bit activations only: the rule is theirs)
  bool decode_fits = c->at != AT_I4 ? (frags <= cus_ || (frags == 2 * cus_ && (m <= 8 || d.K <= 8192 || c->at != AT_F16)) || (m <= 8 && frags <= 2 * cus_))
                                    : (m <= 8 ? frags <= cus_ + cus_ / 2 : (frags <= cus_ && 4 * frags >= 3 * cus_));
  // round 4 - the PERSISTENT form of member 211: where every wave's k-range is one block of 4 k-steps (K <= 32 k-steps: 4096 for the
  // float types, 8192 for int8) a workgroup stages the activations once and takes fragments blk, blk + grid, ... with the next
  // fragments' weights in flight; the grid is capped at one workgroup per CU.  Partial rounds no longer cost a round:
  // 11008 x 4096 M = 3 ... 16 13.3-15.1 us (skinny + reduce) -> 11.6-12.6; 8192 9.8-11.3 -> 8.6-9.7; 5120 9.2-10.6 -> 8.2-9.2
  // (profiles/r04_ab_decode_persistent.txt).  WQAA_GEMM_TUNE=decode_persist=0: off.
  bool persist = false, ksl_ok = false, ksl_take = false, wpf_take = false;
  int ksl_lds = 0, wpf_lds = 0;
  {
    int persist_knob = 1, long_knob = 1, force_knob = -1;     // WQAA_GEMM_TUNE=decode_persist=0 / decode_long=N / decode_force=0|1
    (void)gemm_knob("decode_persist", &persist_knob);
    const bool have_long = gemm_knob("decode_long", &long_knob);
    const bool have_force = gemm_knob("decode_force", &force_knob);
    // (float types, up to three rounds of fragments: 22016 x 4096 - 5.4 per workgroup - and int2 x int8 measured no better
    // than the skinny member + reduce)
    // (the hand-counted form - 4-bit weights, one Scale / Zeros group per k-step - takes up to six rounds of fragments, two batches)
    const bool counted = c->at == AT_F16 && (c->kind == DK_INT4 || c->kind == DK_LUT4) && (c->mode == MD_S || c->mode == MD_ZO || c->mode == MD_ZR) &&
                         g == c->ks && ((d.K / g) & 1) == 0 && d.K / g >= 4;     // (8-byte metadata loads as instructions: 4-byte alignment; four groups per load: a row of two would be read 4 bytes past its end)
    const int pgrid = (cus_ / 8) * 8;                 // the persistent grid (whole XCD rounds): what bounds the fragments per workgroup
    persist = (c->at == AT_F16 || c->at == AT_F8) && frags > cus_ && frags <= (counted ? 6 : 3) * pgrid && nsteps <= 8 * 4 && persist_knob != 0;
    // round 4 - WHOLE TILE, K > 4096 (hand-counted formats): a wave's k-range is nbk = 2 or 3 blocks of 4 k-steps; with slots of
    // nq = ceil(M / 4) KiB per k-step (only the row groups below M) it fits the wave's 16 KiB region for M <= 8 (nbk 2: K <= 8192)
    // and M <= 4 (nbk 3: K <= 12288); units (fragment, block) <= 6 per workgroup.  WQAA_GEMM_TUNE=decode_long=0: off.
    const int run = (((nsteps + 7) / 8) + 3) & ~3, nbk = run / 4, nq = (m + 3) / 4;
    // Taken where there is more than one fragment per workgroup (12288 x 8192 M = 3 / 8: 22.2 / 23.7 us on the skinny member -> 17.7 /
    // 18.0, 8192^2 15.0 / 16.2 -> 12.7 / 13.4, 10240 x 8192 20.2 / 21.2 -> 17.4 / 17.8); with one fragment each (N <= 4096) asking for
    // everything at once measured the same as block by block - 4096 x 11008 M = 4 10.8 vs 11.0 us - and the old form stays
    // (profiles/r04_ab_decode_long.txt).
    const int lmode = have_long ? long_knob : 1;               // 0: none of round 4 / 5's forms; 1: the measured rules; 2: whole tile only (round 4's selector); 3: K-sliced
                                                       // wherever it fits; 4: a wave per fragment wherever it fits
    if (counted && nbk >= 2 && nbk <= 3 && run * nq <= 16 && frags > cus_ && frags <= (6 / nbk) * pgrid && lmode != 0) {
      c->decode_long = 1;
      persist = true;
    }
    // round 5 - K-SLICED, K > 4096 (hand-counted formats, float16): workgroup (k-slice, group) keeps rows < M of ITS eighth of K in LDS
    // (run x nq KiB + the over-read of the last slot) and its waves walk weight fragments; the partial sums (N x 512 B of fp32) meet in
    // wq_mid_reduce_kernel, slices in wave order: the same bits as the one-launch forms.  What it buys: a CU reads an eighth of A once
    // instead of all of A per fragment (at K = 28672 A was twice the weights' bytes per workgroup).  What it costs: the second launch and
    // 128 M / K of the weights' bytes in partial sums - hence long K only.
    ksl_lds = run * nq * 1024 + 4096;
    ksl_ok = !no_mid && counted && !(c->flags & FL_BF16) && nsteps > 32 && ksl_lds <= 160 * 1024 && frags >= 8 && lmode != 0 && lmode != 2 &&
             d.k_split_hint <= 1 && !gemm_knob_set("ksplit") && (long)m * d.K * 2 < (1L << 32) &&
             pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 212) != nullptr;
    // WHERE (same-process A/B against the members it stands in for, uint4 g128 + zeros, us; tools/r05_ab_kslice.py, profiles/r05_ab_kslice.txt):
    // two rounds of fragments or more at K >= 8192 with M = 13 ... 16 - 8192 x 28672 41.1 vs 54.9 (split-K skinny + reduce), 12288 x 8192 22.9
    // vs 26.0, 11008 x 8192 22.4 vs 24.1, 8192^2 18.3 vs 19.2 (M = 9 there: 18.4 vs 17.2, behind) - and M = 5 ... 16 on the longest K (8192 x
    // 28672 M = 8 40.5 vs 46.2; M = 4 a tie).
    // Everywhere else it is BEHIND - 4096 x 11008 M = 8 15.4 vs 11.8, 12288 x 8192 M = 8 22.5 vs 18.0, 4096 x 8192 13.7 vs 8.8: its fixed
    // cost (the request burst of tile + three units per wave at the ~43 GB/s a CU's load path takes, the second launch) is ~9 us against
    // ~3 of the one-launch forms, and only its slope is better (3.8 vs 3.4 TB/s and no second round of A).
    ksl_take = ksl_ok && (lmode == 3 || (frags >= 2 * cus_ && ((m >= 13 && d.K >= 8192) || (m >= 5 && d.K >= 24576))));
    if (ksl_take) {
      c->decode_long = 0;
      persist = true;
    }
    // round 5 - A WAVE PER FRAGMENT, the whole of K (member 213, plan `xdlw`): the same walk with one slice.  The workgroup's waves share
    // the activation tile (nsteps x nq KiB: K <= 4096 at M <= 16, K <= 8192 at M <= 8) and each adds ALL of K for its own fragments in one
    // accumulator and stores them - no meeting, no barrier behind the tile's, no partial sums (other summation order than the forms whose
    // eight waves split K: other bits, same contract).  (lab switch WQAA_GEMM_TUNE=decode_long=4: wherever it fits)
    wpf_lds = ((nsteps + 3) & ~3) * nq * 1024 + 4096;
    // WHERE (tools/r05_ab_wpf.py, profiles/r05_ab_wpf.txt; uint4 g128 + zeros, us): outputs wider than the persistent / whole-tile forms
    // reach, which went to the split-K skinny member + reduce - 32000 x 4096 (a vocabulary projection) M = 4 / 8 / 16 21.3 / 21.4 / 22.6 vs
    // 25.2 / 26.7 / 31.2, 28672 x 4096 20.5 / 20.5 / 21.4 vs 23.4 / 24.7 / 28.2, 16384 x 8192 M = 4 / 8 25.0 / 25.5 vs 27.4 / 28.6.  Its fixed
    // cost (every workgroup stages the whole tile behind one barrier: 12.7 us at 4096^2) keeps it BEHIND wherever those forms exist:
    // 22016 x 4096 19.3 vs 17.3, 11008 x 4096 14.2 vs 9.7, 8192^2 22.5 vs 12.5.
    const bool wpf_ok = !ksl_take && counted && !(c->flags & FL_BF16) && wpf_lds <= 160 * 1024 && frags >= 8 && (long)m * d.K * 2 < (1L << 32) &&
                        pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 213) != nullptr && lmode != 0 && lmode != 2 && lmode != 3;
    // (not where the A/B aids of the older forms are in use: decode_persist=0 / decode_force ask for THOSE members)
    const bool older_forced = persist_knob == 0 || have_force;
    wpf_take = wpf_ok && (lmode == 4 || (!older_forced && !decode_fits && !persist && !c->decode_long && frags > 3 * cus_));
    if (wpf_take) {
      c->decode_long = 0;
      persist = true;
    }
  }
  const bool fits_one_each = decode_fits;     // (the rule for one fragment per workgroup)
  if (persist || c->decode_long) decode_fits = true;
  { int v; if (gemm_knob("decode_force", &v)) decode_fits = v != 0; }   // tuning aid
  if (m <= decode_max_m && m <= 16 && c->mf == 1 && decode_fits) {
    int dflag = 1;
    (void)gemm_knob("decode", &dflag);
    if (dflag != 0) c->decode = 1;
  }
  if (c->decode) {
    // activations through LDS-DMA (member 211); packed int4 activations: the direct-load member 201
    const bool want_lds = c->mf == 1 && c->at != AT_I4;
    c->fn = want_lds ? pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 211) : nullptr;
    const bool lds_member = c->fn != nullptr;
    if (!c->fn) c->fn = pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 200 + c->mf);
    if (c->fn) {
      c->nwaves = 8;
      c->bn = 16;
      c->skinny = 0;
      c->tiles_m = 1;
      c->tiles_n = (d.N + 15) / 16;
      c->lds = lds_member ? 8 * 4 * 16 * 256 + 3 * 8 * 64 * 16 : 8 * c->mf * 64 * 16;     // (three sets of meeting slots: the persistent form)
      c->ksplit = 1;
      c->decode = lds_member ? 2 : 1;
      // persistent: a grid of one workgroup per CU (whole XCD rounds keep the block swizzle on); the direct-load member has no such form
      c->decode_grid = (lds_member && persist) ? (cus_ / 8) * 8 : 0;
      if (!lds_member) c->decode_long = 0;
      gemm_fn ksl_fn = (lds_member && ksl_take) ? pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 212) : nullptr;
      if (ksl_fn) {
        c->fn = ksl_fn;
        // (8 slices x groups of 8 waves: one fragment per wave and round; at most one workgroup per CU - two per CU, where the tile
        // leaves room, measured 5-15 % slower)
        const int groups = (frags + 7) / 8 < cus_ / 8 ? (frags + 7) / 8 : cus_ / 8;
        c->decode_kslice = 1;
        c->decode_grid = 8 * groups;
        c->lds = ksl_lds;
      }
      if (lds_member && wpf_take) {
        c->fn = pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 213);
        int groups = frags < cus_ ? frags : cus_;                      // (fragment = wave x groups + group: every CU gets its share)
        if (groups >= 8) groups = groups / 8 * 8;                     // whole XCD rounds keep the block swizzle on
        c->decode_kslice = 2;
        c->decode_grid = groups;
        c->lds = wpf_lds;
      }
      if (lds_member || fits_one_each) return WQAA_OK;
      c->fn = nullptr;                       // (persistent asked for, but this format has only the direct-load member)
    }
    c->decode = 0;
  }
  // otherwise the skinny member, unless disabled
  const int skinny_max_m = 64;
  c->skinny = (m <= skinny_max_m && c->mf <= 4) ? 4 : 0;
  c->nwaves = c->mf == 16 ? 8 : 4;
  c->bn = c->skinny ? 64 : c->nwaves * 32;
  int code = c->skinny ? 100 + c->mf : c->mf;
  // the 64-row member with Scale AND Zeros, one group per k-step, K / g a multiple of 4: the variant with 8-byte
  // metadata loads (two 2-byte loads per lane and step are as many cache-line touches as the weight load itself;
  // measured same-call: 28672x8192 M=128 143 -> 124 us, 8192^2 51 -> 48 us.  The taller members and the
  // scale-only case LOSE 3-10 % to the four-step unrolled loop, so they keep the per-step form)
  {
    const int dq = g / c->kl;
    int wflag = 1;                                            // WQAA_GEMM_TUNE=wide=0: the per-step metadata loads (test aid)
    (void)gemm_knob("wide", &wflag);
    if (!c->skinny && c->mf == 4 && (c->mode == MD_ZO || c->mode == MD_ZR) && dq == 4 && ((d.K / g) & 3) == 0 &&
        wflag != 0 && pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 404))
      code = 404;
  }
  c->wide = code >= 400;
  c->fn = pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, code);
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
  if (d.k_split_hint > 1) ks = d.k_split_hint;             // the caller's MatmulConfigWithSplitK.k_split
  { int v; if (gemm_knob("ksplit", &v)) ks = v; }
  if (ks > 16) ks = 16;                                     // partial sums cost 4 B per output element per slice
  if (ks > nsteps) ks = nsteps;
  if (ks < 1) ks = 1;
  c->ksplit = ks;
  return WQAA_OK;
}

int gemm_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan, bool fused_epilogue) {
  GemmChoice c;
  int st = gemm_choose(d, m, &c, fused_epilogue);
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
    plan->split_k = (c.mid || c.decode_kslice == 1) ? kMidSlices : c.ksplit;
    plan->lds_bytes = c.lds;
    plan->grid = c.tiles_m * c.tiles_n * (c.mid ? kMidSlices : c.ksplit) + (c.tail_fn ? c.tail_tiles_m * c.tail_tiles_n : 0);
    if (c.decode_grid > 0) plan->grid = c.decode_grid;
    char wd[24];
    short_wdtype(d, wd, sizeof(wd));
    char tail[16] = "";
    if (c.tail_fn) snprintf(tail, sizeof(tail), "t%d", c.tail_tiles_n);      // "ppt11": the last 11 N-tiles as a launch of the 128-row tile
    snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_tcx%dx%dx%d%s%s%s", m, d.N, d.K, short_dtype(d.a_dtype),
             wd, 16 * c.mf, c.bn, c.ks, c.ksplit > 1 ? "xr" : "", c.mid ? "xmk" : c.pp ? "pp" : c.skinny ? "xs" : c.decode == 2 ? (c.decode_kslice == 2 ? "xdlw" : c.decode_kslice ? "xdlk" : c.decode_long ? "xdlt" : c.decode_grid > 0 ? "xdlp" : "xdl") : c.decode ? "xd" : c.wide ? "xw" : "", tail);
  }
  return WQAA_OK;
}

// (the mid-M member's exchange buffer: tiles x 8 portions x 8 slices x mf KiB)
static size_t mid_ws_bytes(const GemmChoice& c) {
  if (c.decode_kslice == 1) return (size_t)c.tiles_n * kMidSlices * 1024;       // (the K-sliced decode form: fragments x 8 slices x 1 KiB)
  return (size_t)c.tiles_m * c.tiles_n * 8 * kMidSlices * c.mf * 1024;
}

size_t gemm_workspace_bytes(const wqaa_matmul_desc& d, int m) {
  GemmChoice c;
  if (gemm_choose(d, m, &c) != WQAA_OK) return 0;
  if (c.mid || c.decode_kslice == 1) {
    // ... or, should its sync words be unavailable at launch time (first use of a device inside a stream capture), the member it
    // stands in for: the larger of the two needs
    GemmChoice f;
    const size_t fb = gemm_choose(d, m, &f, false, true) == WQAA_OK && f.ksplit > 1 ? (size_t)f.ksplit * m * d.N * 4 : 0;
    const size_t mine = mid_ws_bytes(c);
    return mine > fb ? mine : fb;
  }
  return c.ksplit > 1 ? (size_t)c.ksplit * m * d.N * 4 : 0;
}

// the tile map without integer divisions (tile_of_block): reciprocals of its divisors, exact while (largest dividend) x (divisor)
// < 2^32 - beyond that they stay 0 and the kernel divides.  Needs a.tiles_m, a.tiles_n, a.group_m.
static void fill_tile_magics(GemmArgs& a, int ksplit, long K) {
  const unsigned long long ks = (unsigned long long)(ksplit > 0 ? ksplit : 1);
  const unsigned long long blocks = (unsigned long long)a.tiles_m * a.tiles_n * ks;
  const unsigned long long per_group = (unsigned long long)a.group_m * a.tiles_n;
  const int tail = a.tiles_m % a.group_m;
  const bool fits = blocks * (blocks > per_group ? blocks : per_group) < (1ull << 32) && ks * ks * (unsigned long long)K < (1ull << 32);
  a.mg_ntiles = fits ? tile_magic((uint32_t)(a.tiles_m * a.tiles_n)) : 0u;
  a.mg_per_group = fits ? tile_magic((uint32_t)per_group) : 0u;
  a.mg_group_m = fits ? tile_magic((uint32_t)a.group_m) : 0u;
  a.mg_tail_m = fits && tail > 1 ? tile_magic((uint32_t)tail) : 0u;
  a.mg_ksplit = fits ? tile_magic((uint32_t)ks) : 0u;
}

// host twin of the kernels' workgroup -> (k-slice, M-tile, N-tile) map, with the reciprocals the launch would pass (test aid)
void gemm_debug_tile_of_block(int tiles_m, int tiles_n, int ksplit, int group_m, int block, int* out4) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.ksplit = ksplit; a.group_m = group_m;
  fill_tile_magics(a, ksplit, 4096);
  const TileOfBlock t = tile_of_block(a, block, tiles_m * tiles_n * (ksplit > 0 ? ksplit : 1));
  out4[0] = t.split; out4[1] = t.tile_m; out4[2] = t.tile_n;
  out4[3] = (a.mg_ntiles || tiles_m * tiles_n == 1) && (a.mg_per_group || group_m * tiles_n == 1) ? 1 : 0;   // reciprocals in use
}

int gemm_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi, const wqaa_call_opts* opts) {
  GemmChoice c;
  {
    static thread_local ChoiceMemo<GemmChoice> memo;
    const int q = epi ? 1 : 0;            // (with the callers' fused epilogue: another output type, another tile choice)
    if (const GemmChoice* hit = memo.find(d, m, q)) {
      c = *hit;
    } else {
      int st = gemm_choose(d, m, &c, epi != nullptr);
      if (st != WQAA_OK) return st;
      memo.put(d, m, q, c);
    }
  }
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  GemmArgs a;
  // the mid-M member (`xmk`) and the K-sliced decode form (`xdlk`) need a buffer for their slices' partial sums; without it (a
  // short caller workspace, a pool that cannot grow - the first use of a shape inside a stream capture) the call runs the member
  // it stands in for.  That member is chosen once per (descriptor, m) (memoised like the first choice), and the FIRST fallback of
  // a process says so on stderr: `wqaa_select` / the plan name keep reporting the member the shape is planned for, and for `xmk`
  // the stand-in sums K in another order - the output bits of such a call are the stand-in's (include/wqaa.h, wqaa_workspace_bytes:
  // a caller that needs one set of bits eagerly and under capture passes a workspace of that size, or warms the shape up first)
  void* mid_ws = nullptr;
  if ((c.mid || c.decode_kslice == 1) && !epi) {
    const size_t need = mid_ws_bytes(c);
    if (opts && opts->workspace) {
      if (opts->workspace_bytes >= need && (reinterpret_cast<uintptr_t>(opts->workspace) & 15) == 0) mid_ws = opts->workspace;
    } else if (pool_workspace_ready(stream, need)) {
      mid_ws = pool_workspace(stream, need);
    }
    if (!mid_ws) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true))
        fprintf(stderr, "[wqaa] %s: no %zu B of scratch for the slices' partial sums (%s) - this call runs the member it stands in for; "
                "its bits may differ from a call that has the scratch\n", c.mid ? "mid-M member (xmk)" : "K-sliced decode form (xdlk)", need,
                (opts && opts->workspace) ? "caller workspace too small or misaligned" : "the library pool cannot grow during stream capture");
      static thread_local ChoiceMemo<GemmChoice> fallback_memo;
      if (const GemmChoice* hit = fallback_memo.find(d, m, 2)) {
        c = *hit;
      } else {
        int st = gemm_choose(d, m, &c, false, true);
        if (st != WQAA_OK) return st;
        fallback_memo.put(d, m, 2, c);
      }
    }
  }
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
  // "uint8" weights under strict_reference are read as SIGNED storage bytes (see gemv fill_args)
  a.is_signed = d.w_format == WQAA_W_INT || (d.w_format == WQAA_W_UINT && d.w_bits == 8 && d.strict_reference);
  if (d.a_dtype == WQAA_I4) a.is_signed = c.kind == DK_INT4;   // 2-bit weights are zero-extended (matmul_dequantize_mma.py:742-749)
  a.fp4_table = c.fp4_table;
  a.zq_row_bytes = d.N * (c.bits < 8 ? c.bits : 8) / 8;
  // tile order: groups of 4 M-tiles (same-box sweep over 1/2/4/8/16: fp8 4096 x 8192 x 8192 375 -> 361 us,
  // 4096 x 28672 x 8192 1355 -> 1293, uint4 8192^3 1012 -> 984, 4096^3 unchanged; 8 and 16 lose on int2 x int8)
  a.group_m = c.tiles_m >= 4 ? 4 : 1;
  a.tiles_m = c.tiles_m;
  a.tiles_n = c.tiles_n;
  fill_tile_magics(a, c.ksplit, d.K);
  {
    // split-K partial sums and large output tiles leave the chip write-through (they are read by another kernel, once):
    // nothing dirty is left for the kernel boundary to write back.  WQAA_GEMM_TUNE=ws_policy=<bits> overrides (tuning aid, plan time)
    static thread_local unsigned seen = ~0u;
    static thread_local int policy = -1;
    const unsigned ep = g_plan_epoch.load(std::memory_order_relaxed);
    if (ep != seen) {
      int v = -1;
      policy = gemm_knob("ws_policy", &v) ? v : -1;
      seen = ep;
    }
    const long out_bytes = (long)m * d.N * (d.out_dtype == WQAA_I32 || d.out_dtype == WQAA_F32 ? 4 : 2);
    a.ws_policy = policy >= 0 ? policy : (3 | (out_bytes >= (4L << 20) ? 16 : 0));
  }
  a.nsteps = d.K / c.ks;
  a.decode_long = c.decode == 2 ? c.decode_long : 0;
  if (c.pp) {                               // the ping-pong member reads gq_shift as log2(k-bodies per group)
    a.gq_shift = c.pp_shift;
    a.gq_magic = 0u;
  }
  a.epi_row = nullptr;
  a.epi_tensor = 1.f;
  if (epi) {
    if (!at_is_int(c.at) || d.out_dtype != WQAA_F16) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_ex: the fused epilogue needs int8 activations and float16 output");
      return WQAA_ERR_UNSUPPORTED;
    }
    a.epi_row = epi->row_scale;
    a.epi_tensor = epi->tensor_scale;
  }
  a.ksplit = c.ksplit;
  a.ws = nullptr;
  if (c.ksplit > 1) {
    const size_t need = (size_t)c.ksplit * m * d.N * 4;
    if (opts && opts->workspace) {
      if (opts->workspace_bytes < need || (reinterpret_cast<uintptr_t>(opts->workspace) & 15)) {
        set_error(WQAA_ERR_BAD_DESC, "gemm: workspace of %zu B (16-byte aligned) needed, got %zu B at %p", need,
                  (size_t)opts->workspace_bytes, opts->workspace);
        return WQAA_ERR_BAD_DESC;
      }
      a.ws = opts->workspace;
    } else {
      a.ws = pool_workspace(stream, need);
      if (!a.ws) return WQAA_ERR_LAUNCH;
    }
  }
  if (c.decode_kslice == 1) a.ws = mid_ws;
  if (c.mid) {
    a.ws = mid_ws;
    a.mg_ntiles = tile_magic((uint32_t)c.tiles_n);          // (this member's tile map: tile -> (tile / tiles_n, tile % tiles_n))
  }
  void* params[] = {&a};
  dim3 grid(c.decode_grid > 0 ? c.decode_grid : c.tiles_m * c.tiles_n * (c.mid ? kMidSlices : c.ksplit), 1, 1), block(64 * c.nwaves, 1, 1);
  hipError_t e;
  if (start || stop) {
    e = hipExtLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream, start,
                           (c.ksplit > 1 || c.tail_fn || c.mid || c.decode_kslice == 1) ? nullptr : stop, 0);
  } else {
    e = hipLaunchKernel(reinterpret_cast<const void*>(c.fn), grid, block, params, c.lds, stream);
  }
  if (e == hipSuccess && c.tail_fn) {
    // the remainder's columns, by the 128-row tile (same operands; the tile map counts from the first N-tile of the band)
    GemmArgs t = a;
    t.tiles_m = c.tail_tiles_m;
    t.tiles_n = c.tail_tiles_n;
    t.tile_n_off = c.tiles_n;
    t.group_m = a.group_m;
    fill_tile_magics(t, 1, d.K);
    void* tparams[] = {&t};
    const dim3 tgrid(c.tail_tiles_m * c.tail_tiles_n, 1, 1);
    if (start || stop) e = hipExtLaunchKernel(reinterpret_cast<const void*>(c.tail_fn), tgrid, block, tparams, c.tail_lds, stream, nullptr, stop, 0);
    else e = hipLaunchKernel(reinterpret_cast<const void*>(c.tail_fn), tgrid, block, tparams, c.tail_lds, stream);
  }
  if (e == hipSuccess && c.mid) {
    int mfc = c.mf, units = c.tiles_m * c.tiles_n * 8 * c.mf;
    void* rparams[] = {&a, &mfc, &units};
    const dim3 rgrid((unsigned)((units + 3) / 4)), rblock(256);
    const void* rfn = reinterpret_cast<const void*>(wq_mid_reduce_kernel<0>);
    if (start || stop) e = hipExtLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream, nullptr, stop, 0);
    else e = hipLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream);
  }
  if (e == hipSuccess && c.decode_kslice == 1) {
    // unit = fragment: (tile, portion) = (fragment / 8, fragment % 8) of a one-row map of 128-column tiles
    GemmArgs r = a;
    r.tiles_m = 1;
    r.tiles_n = (d.N + 127) / 128;
    r.mg_ntiles = tile_magic((uint32_t)r.tiles_n);
    int mfc = 1, units = c.tiles_n;
    void* rparams[] = {&r, &mfc, &units};
    const dim3 rgrid((unsigned)((units + 3) / 4)), rblock(256);
    const void* rfn = reinterpret_cast<const void*>(wq_mid_reduce_kernel<0>);
    if (start || stop) e = hipExtLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream, nullptr, stop, 0);
    else e = hipLaunchKernel(rfn, rgrid, rblock, rparams, 0, stream);
  }
  if (e == hipSuccess && c.ksplit > 1) {
    const long quads = (long)m * d.N / 4;
    const dim3 rgrid((unsigned)((quads + 255) / 256)), rblock(256);
    const void* ws = a.ws;
    int M_ = m, N_ = d.N, ks_ = c.ksplit, od = d.out_dtype, hb = d.with_bias ? ((c.flags & FL_BF16) ? 2 : 1) : 0;
    const float* er = a.epi_row;
    float et = a.epi_tensor;
    void* rparams[] = {&ws, &C, &Bias, &M_, &N_, &ks_, &od, &hb, &er, &et};
    const void* rfn = !at_is_int(c.at) ? reinterpret_cast<const void*>(wq_splitk_reduce_kernel<true>)
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

// ---- B_decode to memory + the two-pass member -----------------------------------------------------------------------
static int dequant_setup(const wqaa_matmul_desc& d, GemmChoice* c, GemmArgs* a, gemm_fn* fn, const void* B, const void* LUT,
                         const void* Scale, const void* Zeros) {
  int st = gemm_choose(d, 4096, c);              // classification only (kind, layout, activation type, mode, flags)
  if (st != WQAA_OK) return st;
  if (c->at != AT_F16 && c->at != AT_I8) {
    set_error(WQAA_ERR_UNSUPPORTED, "dequantize: B_decode exists for float16 / bfloat16 / int8 activations");
    return WQAA_ERR_UNSUPPORTED;
  }
  *fn = pick_gemm(c->kind, c->layout, c->at, c->mode, c->flags, 900);
  if (!*fn) {
    set_error(WQAA_ERR_UNSUPPORTED, "dequantize: no member for kind=%d layout=%d at=%d mode=%d", c->kind, c->layout, c->at, c->mode);
    return WQAA_ERR_UNSUPPORTED;
  }
  const int g = d.group_size <= 0 ? d.K : d.group_size;
  memset(a, 0, sizeof(*a));
  a->B = B; a->lut = LUT; a->scale = Scale; a->zeros = Zeros;
  a->N = d.N; a->K = d.K;
  a->kg = d.K / g;
  const int dq = g / c->kl > 0 ? g / c->kl : 1;
  a->gq_shift = ilog2_exact(dq);
  a->gq_magic = a->gq_shift >= 0 ? 0u : (uint32_t)(((1ull << 32) + dq - 1) / dq);
  a->row_bytes = (long)d.K * c->bits / 8;
  a->is_signed = d.w_format == WQAA_W_INT || (d.w_format == WQAA_W_UINT && d.w_bits == 8 && d.strict_reference);
  a->fp4_table = c->fp4_table;
  a->zq_row_bytes = d.N * (c->bits < 8 ? c->bits : 8) / 8;
  return WQAA_OK;
}

int gemm_dequantize_launch(const wqaa_matmul_desc& d, const void* B, const void* LUT, const void* Scale, const void* Zeros, void* out,
                           hipStream_t stream) {
  GemmChoice c;
  GemmArgs a;
  gemm_fn fn = nullptr;
  int st = dequant_setup(d, &c, &a, &fn, B, LUT, Scale, Zeros);
  if (st != WQAA_OK) return st;
  const long items = (long)d.N * (d.K / c.kl);
  void* params[] = {&a, &out};
  const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3((unsigned)((items + 255) / 256)), dim3(256), params, 0, stream);
  if (e != hipSuccess) {
    set_error(WQAA_ERR_LAUNCH, "dequantize launch failed: %s", hipGetErrorString(e));
    return WQAA_ERR_LAUNCH;
  }
  return WQAA_OK;
}

// the plain GEMM of the second pass: same N, K, activation / output types, W "stored in A_dtype"
static bool two_pass_dense_desc(const wqaa_matmul_desc& d, wqaa_matmul_desc* dd) {
  if (d.w_format == WQAA_W_NATIVE || d.with_bias) return false;      // nothing to decode / bias after the cast: fused members
  if (d.a_dtype != WQAA_F16 && d.a_dtype != WQAA_BF16 && d.a_dtype != WQAA_I8) return false;
  *dd = d;
  dd->w_format = WQAA_W_NATIVE;
  dd->w_bits = d.a_dtype == WQAA_I8 ? 8 : 16;
  dd->group_size = -1;
  dd->with_scaling = 0;
  dd->zeros_mode = WQAA_Z_NONE;
  dd->w_layout = WQAA_LAYOUT_PLAIN;
  dd->k_split_hint = 0;
  dd->two_pass_min_m = 0;
  return true;
}

// Round 4 - the second pass without the vendor library: this library's own dense 16-bit ping-pong member (PP8Policy<2, 2> / <3, 3>).
// AUTOMATIC for the formats whose fused large-M member is still the lockstep one (float16 / bfloat16 activations x int8, e4m3,
// 2-bit, 1-bit weights ...): B_decode (HBM-bound, ~11 us at 4096^2) + the dense member beats the lockstep fused member from
// ~1024 rows on (profiles/r04_ab_two_pass_own.txt).  Never where a fused ping-pong member exists (it wins: DESIGN.md 3.2a').
static bool own_dense_second_pass(const wqaa_matmul_desc& dd, int m) {
  if (dd.a_dtype != WQAA_F16 && dd.a_dtype != WQAA_BF16 && dd.a_dtype != WQAA_I8) return false;
  GemmChoice c;
  return gemm_choose(dd, m, &c) == WQAA_OK && c.pp;
}

static bool two_pass_auto(const wqaa_matmul_desc& d, const wqaa_matmul_desc& dd, int m) {
  const int auto_m = 1024;
  if (m < auto_m) return false;
  if (d.K % (d.a_dtype == WQAA_I8 ? 256 : 128) != 0 || !own_dense_second_pass(dd, m)) return false;
  // only for descriptors NO fused ping-pong member takes (a format with one keeps its fused members at every shape: where the round
  // estimate prefers the lockstep member - uint4 1024 x 4096^2: 48.9 us - B_decode + dense would be 60)
  GemmChoice c;
  return gemm_choose(d, m, &c) == WQAA_OK && !c.pp_avail;
}

bool gemm_two_pass_eligible(const wqaa_matmul_desc& d, int m) {
  int min_m = d.two_pass_min_m;
  // WQAA_TWO_PASS=min_m=N (or a bare number N): A/B aid (plan-time): 0 never, N > 0 from N rows on
  {
    int v = 0;
    const char* f = getenv("WQAA_TWO_PASS");
    const bool have = f && (knob("WQAA_TWO_PASS", "min_m", &v) || ((*f >= '0' && *f <= '9') && ((v = atoi(f)), true)));
    if (have) {
      min_m = v;
      if (min_m <= 0) return false;
    }
  }
  wqaa_matmul_desc dd;
  if (m < 16 || !two_pass_dense_desc(d, &dd)) return false;
  GemmChoice c;
  GemmArgs a;
  gemm_fn fn = nullptr;
  if (dequant_setup(d, &c, &a, &fn, nullptr, nullptr, nullptr, nullptr) != WQAA_OK) return false;
  const bool asked = min_m > 0 && m >= min_m;
  if (asked && (dense_lib_eligible(dd, m, true) || own_dense_second_pass(dd, m))) return true;
  return two_pass_auto(d, dd, m);
}

// the vendor GEMM of the second pass, tuned (dense_lib_tune) - independent of two_pass_min_m, which the caller sets afterwards
int gemm_two_pass_tune(const wqaa_matmul_desc& d, int m, hipStream_t stream) {
  wqaa_matmul_desc dd;
  if (m < 16 || !two_pass_dense_desc(d, &dd) || !dense_lib_eligible(dd, m, true)) return WQAA_OK;     // nothing to tune
  return dense_lib_tune(dd, m, stream, nullptr);
}

static size_t two_pass_scratch(const wqaa_matmul_desc& d) {
  const size_t esz = d.a_dtype == WQAA_I8 ? 1 : 2;
  return ((size_t)d.N * d.K * esz + 255) & ~(size_t)255;
}

size_t gemm_two_pass_workspace_bytes(const wqaa_matmul_desc& d, int m) {
  wqaa_matmul_desc dd;
  if (!two_pass_dense_desc(d, &dd)) return 0;
  return two_pass_scratch(d) + (dense_lib_eligible(dd, m, true) ? dense_lib_workspace_bytes(dd, m) : 0);
}

int gemm_two_pass_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan) {
  wqaa_matmul_desc dd;
  if (!two_pass_dense_desc(d, &dd)) {
    set_error(WQAA_ERR_UNSUPPORTED, "two-pass member: not defined for this configuration");
    return WQAA_ERR_UNSUPPORTED;
  }
  const bool vendor = dense_lib_eligible(dd, m, true);
  int st = vendor ? dense_lib_plan(dd, m, plan) : gemm_plan(dd, m, plan);
  if (st == WQAA_OK && plan) {
    plan->kernel_family = 4;
    char wd[24];
    short_wdtype(d, wd, sizeof(wd));
    if (vendor) {
      snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_dq_hipblaslt", m, d.N, d.K, short_dtype(d.a_dtype), wd);
    } else {
      // "..._dq_tcx256x256x64pp": B_decode, then the dense member of that tile
      char tile[48];
      const char* t = strstr(plan->name, "_tcx");
      snprintf(tile, sizeof(tile), "%s", t ? t + 1 : "own");
      snprintf(plan->name, sizeof(plan->name), "matmul_m%dn%dk%d_%sx%s_dq_%s", m, d.N, d.K, short_dtype(d.a_dtype), wd, tile);
    }
  }
  return st;
}

int gemm_two_pass_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT, const void* Scale, const void* Zeros,
                         void* C, int m, hipStream_t stream, const wqaa_call_opts* opts) {
  wqaa_matmul_desc dd;
  if (!two_pass_dense_desc(d, &dd)) {
    set_error(WQAA_ERR_UNSUPPORTED, "two-pass member: not defined for this configuration");
    return WQAA_ERR_UNSUPPORTED;
  }
  const bool vendor = dense_lib_eligible(dd, m, true);
  const size_t scratch = two_pass_scratch(d);
  const size_t need = scratch + (vendor ? dense_lib_workspace_bytes(dd, m) : 0);
  uint8_t* ws = nullptr;
  if (opts && opts->workspace) {
    if (opts->workspace_bytes < need || (reinterpret_cast<uintptr_t>(opts->workspace) & 15)) {
      set_error(WQAA_ERR_BAD_DESC, "two-pass member: workspace of %zu B (16-byte aligned) needed, got %zu B at %p", need,
                (size_t)opts->workspace_bytes, opts->workspace);
      return WQAA_ERR_BAD_DESC;
    }
    ws = reinterpret_cast<uint8_t*>(opts->workspace);
  } else {
    ws = reinterpret_cast<uint8_t*>(pool_workspace(stream, need));
    if (!ws) return WQAA_ERR_LAUNCH;
  }
  int st = gemm_dequantize_launch(d, B, LUT, Scale, Zeros, ws, stream);
  if (st != WQAA_OK) return st;
  if (!vendor) return gemm_launch(dd, A, ws, nullptr, nullptr, nullptr, nullptr, C, m, stream, nullptr, nullptr, nullptr, nullptr);
  wqaa_call_opts sub;
  memset(&sub, 0, sizeof(sub));
  sub.struct_size = (int32_t)sizeof(sub);
  sub.workspace = need > scratch ? ws + scratch : nullptr;
  sub.workspace_bytes = need - scratch;
  return dense_lib_launch(dd, A, ws, C, m, stream, need > scratch ? &sub : nullptr);
}

void gemm_init() {
  const int kinds[] = {DK_INT4, DK_INT2, DK_INT1, DK_INT8, DK_LUT4, DK_E4M3, DK_E5M2, DK_NATIVE};
  for (int kind : kinds)
    for (int layout = 0; layout < 2; ++layout)
      for (int at = 0; at < 4; ++at)
        for (int mode = 0; mode <= MD_ZQ; ++mode)
          for (int flags : {0, (int)FL_STRICT, (int)FL_ABF8, (int)FL_BF16})
            for (int mf : {4, 8, 16, 101, 102, 104, 201, 211, 212, 213, 404}) {
              gemm_fn fn = pick_gemm(kind, layout, at, mode, flags, mf);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
  for (int kind : {DK_INT4, DK_LUT4, DK_INT2, DK_E4M3, DK_E5M2})
    for (int layout = 0; layout < 2; ++layout)
      for (int at : {AT_F16, AT_I8, AT_F8})
        for (int mode = 0; mode <= MD_ZQ; ++mode)
          for (int flags : {0, (int)FL_ABF8, (int)FL_BF16}) {
          for (int bm : {256, 128})
            for (int bn : {256, 128}) {
              int lds = 0;
              gemm_fn fn = pick_gemm_pp(kind, layout, at, mode, flags, bm, bn, &lds);
              if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
        }
  for (int layout = 0; layout < 2; ++layout)
    for (int mode = 0; mode <= MD_ZQ; ++mode)
      for (int mf : {2, 4, 8})
        for (int nkh : {2, 4}) {
          int lds = 0;
          gemm_fn fn = pick_gemm_mid(DK_INT4, layout, mode, mf, nkh, &lds);
          if (fn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
  (void)hipGetLastError();   // a refused attribute must not linger as this thread's "last error"
}

}  // namespace wqaa
