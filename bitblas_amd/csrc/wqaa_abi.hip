// wqaa_abi.hip - extern "C" entry points of libwqaa_hip.so (see include/wqaa.h).
//
// Replaces the reference's generated per-config wrapper (`init()` + `call()`,
// bitblas/builder/wrapper/tl.py:90-166 and :200-305) and its host-side kernel choice
// (`MatmulDequantizeScheduler.dispatch_*`, tilelang/dequantize/matmul_dequantize.py:65-155:
// M < 8 -> GEMV, otherwise the tensor-core GEMM; here the switch sits at M = 3, see dispatch()).
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "wqaa_common.h"
#include "wqaa_kinds.h"

namespace wqaa {

std::atomic<unsigned> g_plan_epoch{1};
static thread_local int g_last_error = WQAA_OK;
static thread_local char g_last_error_msg[512] = "";

void set_error(int code, const char* fmt, ...) {
  g_last_error = code;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error_msg, sizeof(g_last_error_msg), fmt, ap);
  va_end(ap);
}

// ---- devices ---------------------------------------------------------------------------------
// Everything device-specific is per device and lazy: the CU count the selector sizes grids with, and the
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) calls of gemv_init / gemm_init (function attributes are per
// device).  A launch runs on the device that OWNS ITS STREAM (hipStreamGetDevice), made current for the duration of
// the call when it is not - so `A` on cuda:1 with cuda:0 current works without a device guard in the caller
// (the reference leaves this to torch's current device; upstream wrapper: bitblas/builder/wrapper/tl.py:104-120).
static constexpr int kMaxDevices = 32;
static DeviceInfo g_dev[kMaxDevices];
static std::once_flag g_dev_once[kMaxDevices];
static const DeviceInfo g_no_dev = {0, 256, 160 * 1024, "gfx950"};
static int g_ndev = -1;
static std::once_flag g_ndev_once;

static int device_count_cached() {
  std::call_once(g_ndev_once, [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 0;
    }
    g_ndev = n < kMaxDevices ? n : kMaxDevices;
  });
  return g_ndev;
}

// properties of device `dev` + the per-device kernel attributes, once (the device must be current)
static void ensure_device(int dev) {
  if (dev < 0 || dev >= device_count_cached()) return;
  std::call_once(g_dev_once[dev], [dev] {
    DeviceInfo& di = g_dev[dev];
    di = g_no_dev;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
      di.cus = p.multiProcessorCount;
      di.lds_per_block = (int)p.maxSharedMemoryPerMultiProcessor;
      snprintf(di.arch, sizeof(di.arch), "%s", p.gcnArchName);
      di.ok = 1;
      gemv_init();
      gemm_init();
    } else {
      (void)hipGetLastError();
    }
  });
}

const DeviceInfo& device_info() {
  if (device_count_cached() <= 0) return g_no_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= g_ndev) {
    (void)hipGetLastError();
    return g_no_dev;
  }
  ensure_device(dev);
  return g_dev[dev];
}

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return dev;
}

// RAII: make the launch device current (and initialised) for one call.  The device is the stream's; the NULL stream exists
// on every device (torch's default stream is handle 0 everywhere), so there the device is taken from the operand `ptr`
// (device memory owned by the caller): A on cuda:1 with cuda:0 current launches on cuda:1's null stream, not cuda:0's.
struct StreamDeviceScope {
  int prev = -1, dev = -1;
  bool switched = false;
  explicit StreamDeviceScope(hipStream_t s, const void* ptr = nullptr) {
    if (device_count_cached() <= 0) return;
    (void)hipGetDevice(&prev);
    dev = prev;
    if (g_ndev > 1) {
      if (s != nullptr) {
        hipDevice_t sd = 0;
        if (hipStreamGetDevice(s, &sd) == hipSuccess) dev = (int)sd;
        else (void)hipGetLastError();
      } else if (ptr != nullptr) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice && at.device >= 0 && at.device < g_ndev) dev = at.device;
        else (void)hipGetLastError();
      }
    }
    if (dev != prev) switched = hipSetDevice(dev) == hipSuccess;
    ensure_device(dev);
  }
  ~StreamDeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static bool valid_desc(const wqaa_matmul_desc* d) {
  if (!d) {
    set_error(WQAA_ERR_BAD_DESC, "null descriptor");
    return false;
  }
  if (d->struct_size != (int32_t)sizeof(wqaa_matmul_desc)) {
    set_error(WQAA_ERR_BAD_DESC, "descriptor size %d != %zu (ABI mismatch)", d->struct_size, sizeof(wqaa_matmul_desc));
    return false;
  }
  if (d->N <= 0 || d->K <= 0) {
    set_error(WQAA_ERR_BAD_DESC, "N=%d K=%d must be positive", d->N, d->K);
    return false;
  }
  // packed zero points: a row of Qzeros holds N fields of w_bits in whole bytes (matmul_dequantize_impl.py:375-389 sizes it
  // N // 8 * bit and reads past it otherwise)
  if (d->zeros_mode == WQAA_Z_QUANTIZED && d->w_bits > 0 && d->w_bits < 8 && ((long)d->N * d->w_bits) % 8 != 0) {
    set_error(WQAA_ERR_BAD_DESC, "quantized zero points: N=%d x %d bits must fill whole bytes", d->N, d->w_bits);
    return false;
  }
  return true;
}

// M <= 2 -> GEMV family; larger m -> MFMA GEMM when a member exists for the dtype pair and shape, else the
// GEMV family iterates over batch tiles of 4 rows.
static int dispatch(const wqaa_matmul_desc& d, int m, bool* use_gemm) {
  *use_gemm = false;
  wqaa_plan p;
  const int saved = g_last_error;
  char saved_msg[sizeof(g_last_error_msg)];
  memcpy(saved_msg, g_last_error_msg, sizeof(saved_msg));
  // The reference switches families at M = 8 (matmul_dequantize.py:93-102).  Measured here (same box,
  // int4 g128, N x K): the 4-row GEMV batch tile pays 4 LDS reads + 16 dot2 per weight word and, for
  // M = 5..7, streams the weights twice, while one MFMA covers 16 rows:
  //   M = 3   4096^2 7.5 vs 6.4 us (decode-batch MFMA member), 4096 x 11008 21.4 vs 11.7, int2 x int8 6.7 vs 5.5
  //   M = 4   4096^2 7.8 vs 6.6,  11008 x 4096 14.4 vs 13.2 (skinny member + reduce launch)
  //   M = 5-7 4096^2 10.9-11.8 vs 7.6-8.2
  // M <= 2 stays on the GEMV family (M = 2: 5.2 us at 4096^2, the MFMA members take ~6.2).
  const int min_m = 3;
  if (m >= min_m) {
    if (gemm_plan(d, m, &p) == WQAA_OK) *use_gemm = true;
  } else if (gemv_plan(d, m, &p) != WQAA_OK && gemm_plan(d, m, &p) == WQAA_OK) {
    // the GEMV family refuses this config (e.g. groups smaller than its 16-byte lane chunk) but the
    // MFMA family's skinny member covers it: correct first, the selector's preference second
    *use_gemm = true;
  }
  g_last_error = saved;
  memcpy(g_last_error_msg, saved_msg, sizeof(saved_msg));
  return WQAA_OK;
}

// WQAA_TWO_PASS (A/B aid: the two-pass member wherever it exists) is a plan-time switch like every WQAA_* variable: read when
// wqaa_select / a plan query bumps the epoch, not on every launch
static bool two_pass_forced() {
  static thread_local unsigned seen_epoch = ~0u;
  static thread_local bool on = false;
  const unsigned ep = g_plan_epoch.load(std::memory_order_relaxed);
  if (ep != seen_epoch) {
    int v;
    const char* f = getenv("WQAA_TWO_PASS");
    on = f && (knob("WQAA_TWO_PASS", "min_m", &v) || (*f >= '0' && *f <= '9'));
    seen_epoch = ep;
  }

  return on;
}

// The two-pass member as the PLAN sees it: eligible, and - for the automatic form (no caller threshold, not forced) - its scratch
// N K sizeof(A_dtype) within the cap (WQAA_TWO_PASS=auto_max_mb=N, default 256).  wqaa_select, wqaa_workspace_bytes and the call agree.
static size_t two_pass_auto_cap() {
  int mb = 256;                                  // WQAA_TWO_PASS=auto_max_mb=N
  (void)knob("WQAA_TWO_PASS", "auto_max_mb", &mb);
  return (size_t)(mb < 0 ? 0 : mb) << 20;
}
static bool two_pass_planned(const wqaa_matmul_desc& d, int m) {
  if (!gemm_two_pass_eligible(d, m)) return false;
  const bool automatic = d.two_pass_min_m <= 0 && !two_pass_forced();
  return !automatic || gemm_two_pass_workspace_bytes(d, m) <= two_pass_auto_cap();
}

constexpr int32_t kEpilogueV1Bytes = 24;   // wqaa_epilogue up to `reserved2`: callers built before the float16 pre/post ops

static int matmul_impl(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                       const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                       void* stream, void* ev0, void* ev1, const wqaa_epilogue* epi = nullptr,
                       const wqaa_call_opts* opts = nullptr) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  if (m == 0) return WQAA_OK;  // wrapper/tl.py:277
  if (m < 0 || !A || !B || !C) {
    set_error(WQAA_ERR_BAD_DESC, "bad call: m=%d A=%p B=%p C=%p", m, A, B, C);
    return WQAA_ERR_BAD_DESC;
  }
  if (desc->with_scaling && !Scale) {
    set_error(WQAA_ERR_BAD_DESC, "with_scaling set but Scale is NULL");
    return WQAA_ERR_BAD_DESC;
  }
  if (desc->zeros_mode != WQAA_Z_NONE && !Zeros) {
    set_error(WQAA_ERR_BAD_DESC, "zeros_mode=%d but Zeros is NULL", desc->zeros_mode);
    return WQAA_ERR_BAD_DESC;
  }
  if (desc->with_bias && !Bias) {
    set_error(WQAA_ERR_BAD_DESC, "with_bias set but Bias is NULL");
    return WQAA_ERR_BAD_DESC;
  }
  if (desc->w_format == WQAA_W_NF && !LUT) {
    set_error(WQAA_ERR_BAD_DESC, "nf weights need the LUT pointer");
    return WQAA_ERR_BAD_DESC;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  StreamDeviceScope scope(s, A);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  bool use_gemm = false;
  {
    static thread_local ChoiceMemo<int> memo;   // family per (desc, m), see ChoiceMemo
    if (const int* hit = memo.find(*desc, m, 0)) {
      use_gemm = *hit != 0;
    } else {
      dispatch(*desc, m, &use_gemm);
      memo.put(*desc, m, 0, use_gemm ? 1 : 0);
    }
  }
  hipEvent_t e0 = reinterpret_cast<hipEvent_t>(ev0), e1 = reinterpret_cast<hipEvent_t>(ev1);
  // plain dense GEMMs (no decode, no bias, no caller epilogue) at M >= 16: the vendor library (wqaa_dense_lib.hip)
  if (!epi && !e0 && !e1) {
    static thread_local ChoiceMemo<int> lib_memo;
    bool lib;
    if (const int* hit = lib_memo.find(*desc, m, 3)) {
      lib = *hit != 0;
    } else {
      const int saved = g_last_error;
      lib = dense_lib_eligible(*desc, m);
      g_last_error = saved;
      lib_memo.put(*desc, m, 3, lib ? 1 : 0);
    }
    if (lib) {
      int st = dense_lib_launch(*desc, A, B, C, m, s, opts);
      if (st == WQAA_OK) g_last_error = WQAA_OK;
      return st;
    }
  }
  // the two-pass member: B_decode to a scratch, then the plain GEMM (desc.two_pass_min_m: the caller's tuned threshold; m >= 256: the
  // automatic form - own B_decode + own dense member where the fused member is still a lockstep one).  Also under wqaa_matmul_timed
  // (events recorded around the two launches): what is timed is what is dispatched.
  if (!epi && (desc->two_pass_min_m > 0 || two_pass_forced() || m >= 256)) {
    static thread_local ChoiceMemo<int> tp_memo;
    bool tp;
    if (const int* hit = tp_memo.find(*desc, m, 5)) {
      tp = *hit != 0;
    } else {
      const int saved = g_last_error;
      tp = gemm_two_pass_eligible(*desc, m);
      g_last_error = saved;
      tp_memo.put(*desc, m, 5, tp ? 1 : 0);
    }
    const bool automatic = desc->two_pass_min_m <= 0 && !two_pass_forced();
    if (tp && automatic) {
      // the AUTOMATIC form never turns a call that used to need no scratch into a refused or a memory-hungry one:
      //  * its scratch is N K sizeof(A_dtype) per stream - capped (WQAA_TWO_PASS=auto_max_mb=N, default 256: a 8192 x 28672 layer
      //    would pin 470 MB per stream to save ~15 % of a prefill call);
      //  * a caller workspace too small for it (sized for the fused member's needs: none), or a library pool that cannot grow
      //    (stream capture), means the fused member runs;
      //  * and so does a failed allocation (below).
      static thread_local unsigned seen = ~0u;
      static thread_local size_t cap = (size_t)256 << 20;
      const unsigned ep = g_plan_epoch.load(std::memory_order_relaxed);
      if (ep != seen) {
        cap = two_pass_auto_cap();
        seen = ep;
      }
      const size_t need = gemm_two_pass_workspace_bytes(*desc, m);
      if (need > cap) tp = false;
      else if (opts && opts->workspace) {
        if (opts->workspace_bytes < need || (reinterpret_cast<uintptr_t>(opts->workspace) & 15)) tp = false;
      } else if (!pool_workspace_ready(s, need)) tp = false;
    }
    if (tp) {
      if (e0 && hipEventRecord(e0, s) != hipSuccess) (void)hipGetLastError();
      const int saved = g_last_error;
      int st = gemm_two_pass_launch(*desc, A, B, LUT, Scale, Zeros, C, m, s, opts);
      if (st == WQAA_OK) {
        if (e1 && hipEventRecord(e1, s) != hipSuccess) (void)hipGetLastError();
        g_last_error = WQAA_OK;
        return st;
      }
      if (!automatic) return st;
      g_last_error = saved;              // (the automatic form failed to get its scratch: the fused member, as before round 4)
    }
  }
  const bool quant_in = epi && (epi->flags & WQAA_EPI_QUANTIZE_INPUT);
  const bool float_ops = epi && (epi->flags & (WQAA_EPI_ADD_RESIDUAL | WQAA_EPI_RMSNORM_INPUT));
  if (float_ops) {
    // the float16 path's residual add / norm in front: the whole descriptor, no int8 flags, the pointers present
    if (epi->struct_size != (int32_t)sizeof(wqaa_epilogue) || quant_in || ((epi->flags & WQAA_EPI_ADD_RESIDUAL) && !epi->residual) ||
        ((epi->flags & WQAA_EPI_RMSNORM_INPUT) && !epi->norm_weight)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_ex: malformed epilogue descriptor (residual add / RMSNorm input)");
      return WQAA_ERR_BAD_DESC;
    }
    if ((epi->flags & WQAA_EPI_ADD_RESIDUAL) && (epi->flags & WQAA_EPI_RMSNORM_INPUT)) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_ex: RMSNorm input and residual add on one launch (no layer has the two on one projection)");
      return WQAA_ERR_UNSUPPORTED;
    }
    use_gemm = false;   // an exact-product GEMV member; refuses loudly if the config has none
  } else if (epi && ((epi->struct_size != (int32_t)sizeof(wqaa_epilogue) && epi->struct_size != kEpilogueV1Bytes) ||
                     (!epi->row_scale && !quant_in))) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_ex: malformed epilogue descriptor");
    return WQAA_ERR_BAD_DESC;
  }
  if (quant_in) {
    if (m > 4) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_ex: in-kernel activation quantisation covers m <= 4 (got %d)", m);
      return WQAA_ERR_UNSUPPORTED;
    }
    use_gemm = false;   // a GEMV-family member; refuses loudly if the config has none
  }
  int st = use_gemm ? gemm_launch(*desc, A, B, LUT, Scale, Zeros, Bias, C, m, s, e0, e1, epi, opts)
                    : gemv_launch(*desc, A, B, LUT, Scale, Zeros, Bias, C, m, s, e0, e1, epi);
  if (st == WQAA_OK) g_last_error = WQAA_OK;
  return st;
}

// ---- groups (wqaa_matmul_group) ------------------------------------------------------------------------------------
// A group fuses into one launch when every member has the same descriptor apart from N and the merged operator
// (N = the sum of the members' rows) is served by a GEMV-family member at this m.  *fused_x: 1 = exact-product family.
static bool group_fusable(const wqaa_matmul_desc* const* descs, int count, int m, wqaa_matmul_desc* merged, int* fused_x,
                          int epi_mode = 0) {      // epi_mode: 0 none, 1 caller's row scales, 2 in-kernel activation quantiser, 3 RMSNorm in front
  if (count < 2 || count > WQAA_GROUP_MAX || m < 1 || m > 2) return false;
  long total = 0;
  for (int i = 0; i < count; ++i) {
    wqaa_matmul_desc a = *descs[i], b = *descs[0];
    a.N = b.N = 0;
    if (memcmp(&a, &b, sizeof(a)) != 0) return false;
    total += descs[i]->N;
  }
  if (total > 0x7fffffffl) return false;
  *merged = *descs[0];
  merged->N = (int32_t)total;
  bool use_gemm = false;
  dispatch(*merged, m, &use_gemm);
  if (use_gemm && epi_mode != 2) return false;
  const int saved = g_last_error;
  char saved_msg[sizeof(g_last_error_msg)];
  memcpy(saved_msg, g_last_error_msg, sizeof(saved_msg));
  bool ok = true;
  if (epi_mode == 3) {
    if (gemvx_group_eligible(*merged, descs, count, m, true)) *fused_x = 1;
    else ok = false;
  } else if (epi_mode == 0 && gemvx_group_eligible(*merged, descs, count, m)) *fused_x = 1;
  else if (gemv_group_eligible(*merged, descs, count, m, epi_mode != 0, epi_mode == 2)) {
    *fused_x = 0;
    // a member whose OWN call takes the exact-product family (refused above because its K split alone differs from the merged
    // operator's: 4096 x 8192 alone splits K in two, 3 x 4096 merged does not) must not be fused into the rounding family's
    // launch: other arithmetic, other bits than `wqaa_matmul` on it (found by tests/test_group_gpu.py, round 4)
    for (int i = 0; epi_mode == 0 && descs && i < count; ++i)
      if (gemvx_eligible(*descs[i], m)) ok = false;
  } else ok = false;
  g_last_error = saved;
  memcpy(g_last_error_msg, saved_msg, sizeof(saved_msg));
  return ok;
}

}  // namespace wqaa

using namespace wqaa;

extern "C" {

void init(void) {
  static std::once_flag once;
  std::call_once(once, [] { (void)device_info(); });   // current device now; any other device at its first launch
}

int wqaa_abi_version(void) { return WQAA_ABI_VERSION; }

int wqaa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int wqaa_matmul(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream) {
  return matmul_impl(desc, A, B, LUT, Scale, Zeros, Bias, C, m, stream, nullptr, nullptr);
}

int wqaa_matmul_timed(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                      const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                      void* stream, void* start_event, void* stop_event) {
  return matmul_impl(desc, A, B, LUT, Scale, Zeros, Bias, C, m, stream, start_event, stop_event);
}

int wqaa_matmul_ex(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                   const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream,
                   const wqaa_epilogue* epilogue) {
  return matmul_impl(desc, A, B, LUT, Scale, Zeros, Bias, C, m, stream, nullptr, nullptr, epilogue);
}

uint64_t wqaa_workspace_bytes(const wqaa_matmul_desc* desc, int m) {
  if (!valid_desc(desc) || m <= 0) return 0;
  if (dense_lib_eligible(*desc, m)) return (uint64_t)dense_lib_workspace_bytes(*desc, m);
  if (two_pass_planned(*desc, m)) return (uint64_t)gemm_two_pass_workspace_bytes(*desc, m);
  bool use_gemm = false;
  dispatch(*desc, m, &use_gemm);
  return use_gemm ? (uint64_t)gemm_workspace_bytes(*desc, m) : 0;
}

int wqaa_matmul_opts(const wqaa_matmul_desc* desc, const void* A, const void* B, const void* LUT,
                     const void* Scale, const void* Zeros, const void* Bias, void* C, int m, void* stream,
                     const wqaa_call_opts* opts) {
  if (opts && opts->struct_size != (int32_t)sizeof(wqaa_call_opts)) {
    set_error(WQAA_ERR_BAD_DESC, "call options size %d != %zu (ABI mismatch)", opts->struct_size, sizeof(wqaa_call_opts));
    return WQAA_ERR_BAD_DESC;
  }
  return matmul_impl(desc, A, B, LUT, Scale, Zeros, Bias, C, m, stream, nullptr, nullptr, opts ? opts->epilogue : nullptr, opts);
}

int wqaa_group_plan(const wqaa_matmul_desc* const* descs, int count, int m, int* launches, wqaa_plan* plan) {
  if (!descs || count < 1) {
    set_error(WQAA_ERR_BAD_DESC, "group_plan: no members");
    return WQAA_ERR_BAD_DESC;
  }
  for (int i = 0; i < count; ++i)
    if (!valid_desc(descs[i])) return WQAA_ERR_BAD_DESC;
  if (plan) memset(plan, 0, sizeof(*plan));
  if (m <= 0) m = 1;
  g_plan_epoch.fetch_add(1, std::memory_order_relaxed);
  wqaa_matmul_desc merged;
  int fx = 0;
  if (!group_fusable(descs, count, m, &merged, &fx)) {
    if (launches) *launches = count;
    return WQAA_OK;                      // members run one by one: their own plans are wqaa_select's
  }
  if (launches) *launches = 1;
  int Ns[WQAA_GROUP_MAX];
  for (int i = 0; i < count; ++i) Ns[i] = descs[i]->N;
  return fx ? gemvx_group_plan(merged, Ns, count, m, plan) : gemv_group_plan(merged, Ns, count, m, plan);
}

static int group_impl(const wqaa_group_item* items, const wqaa_epilogue* const* epis, int count, int m, void* stream) {
  if (count == 0 || m == 0) return WQAA_OK;       // an empty group / m == 0 returns at once (wrapper/tl.py:277)
  if (!items || count < 0) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_group: bad arguments (items=%p count=%d)", (const void*)items, count);
    return WQAA_ERR_BAD_DESC;
  }
  // epilogues: all members or none; the fused launch needs one kind (caller's row scales / in-kernel quantiser)
  int epi_mode = 0;
  bool epi_uniform = true;
  if (epis) {
    for (int i = 0; i < count; ++i) {
      const wqaa_epilogue* e = epis[i];
      const bool norm = e && e->struct_size == (int32_t)sizeof(wqaa_epilogue) && e->flags == WQAA_EPI_RMSNORM_INPUT && e->norm_weight;
      if (!norm && (!e || (e->struct_size != (int32_t)sizeof(wqaa_epilogue) && e->struct_size != kEpilogueV1Bytes) ||
                    (e->flags & (WQAA_EPI_ADD_RESIDUAL | WQAA_EPI_RMSNORM_INPUT)) || (!e->row_scale && !(e->flags & WQAA_EPI_QUANTIZE_INPUT)))) {
        set_error(WQAA_ERR_BAD_DESC, "matmul_group_ex: member %d has a missing or malformed epilogue descriptor", i);
        return WQAA_ERR_BAD_DESC;
      }
      const int mode = norm ? 3 : (e->flags & WQAA_EPI_QUANTIZE_INPUT) ? 2 : 1;
      if (i == 0) epi_mode = mode;
      else if (mode != epi_mode) epi_uniform = false;
      // one norm per fused group: its members read the same hidden state through the same weight
      if (norm && (e->norm_weight != epis[0]->norm_weight || e->norm_eps != epis[0]->norm_eps || items[i].A != items[0].A)) epi_uniform = false;
    }
  }
  const wqaa_matmul_desc* descs[WQAA_GROUP_MAX];
  bool fuse = count >= 2 && count <= WQAA_GROUP_MAX && m >= 1 && m <= 2 && epi_uniform;
  for (int i = 0; i < count && fuse; ++i) {
    const wqaa_group_item& it = items[i];
    if (!valid_desc(it.desc)) return WQAA_ERR_BAD_DESC;
    // the per-member pointer checks of wqaa_matmul; a malformed member fails the whole group before anything is launched
    if (!it.A || !it.B || !it.C || (it.desc->with_scaling && !it.Scale) || (it.desc->zeros_mode != WQAA_Z_NONE && !it.Zeros) ||
        (it.desc->with_bias && !it.Bias) || (it.desc->w_format == WQAA_W_NF && !it.LUT)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_group: member %d has a null operand its descriptor requires", i);
      return WQAA_ERR_BAD_DESC;
    }
    descs[i] = it.desc;
  }
  if (fuse) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StreamDeviceScope scope(s, items[0].A);
    if (!device_info().ok) {
      set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
      return WQAA_ERR_NO_DEVICE;
    }
    wqaa_matmul_desc merged;
    int fx = 0;
    bool ok;
    {
      // fusability per (merged descriptor, m, count, epilogue kind): memoised like every tile choice
      struct Fuse { int ok, fx; };
      static thread_local ChoiceMemo<Fuse> memo;
      wqaa_matmul_desc key = *descs[0];
      long total = 0;
      bool same = true;
      for (int i = 0; i < count; ++i) {
        wqaa_matmul_desc a = *descs[i], b = *descs[0];
        a.N = b.N = 0;
        same = same && memcmp(&a, &b, sizeof(a)) == 0;
        total += descs[i]->N;
      }
      key.N = (int32_t)(total & 0x7fffffff);
      // fusability depends on every member's OWN N (each alone must land on the merged K split / family): two groups of one
      // total and count but other splits (3 x 4096 vs 8192 + 2 x 2048) must not share a verdict - the split goes into the key
      uint32_t nhash = 2166136261u;
      for (int i = 0; i < count; ++i) nhash = (nhash ^ (uint32_t)descs[i]->N) * 16777619u;
      key.reserved[0] = (int32_t)nhash;
      const int q = 32 + count + 16 * epi_mode;
      const Fuse* hit = same ? memo.find(key, m, q) : nullptr;
      if (hit) {
        ok = hit->ok != 0;
        fx = hit->fx;
        merged = key;
        merged.reserved[0] = 0;
      } else {
        ok = group_fusable(descs, count, m, &merged, &fx, epi_mode);
        if (same) memo.put(key, m, q, Fuse{ok ? 1 : 0, fx});
      }
    }
    if (ok) {
      int st = fx ? gemvx_group_launch(merged, items, count, m, s, epi_mode == 3 ? epis[0] : nullptr)
                  : gemv_group_launch(merged, items, count, m, s, epis);
      if (st == WQAA_OK) g_last_error = WQAA_OK;
      return st;
    }
  }
  for (int i = 0; i < count; ++i) {
    const wqaa_group_item& it = items[i];
    int st = matmul_impl(it.desc, it.A, it.B, it.LUT, it.Scale, it.Zeros, it.Bias, it.C, m, stream, nullptr, nullptr,
                         epis ? epis[i] : nullptr);
    if (st != WQAA_OK) return st;
  }
  return WQAA_OK;
}

int wqaa_matmul_group(const wqaa_group_item* items, int count, int m, void* stream) {
  return group_impl(items, nullptr, count, m, stream);
}

int wqaa_matmul_gate_up(const wqaa_group_item* gate, const wqaa_group_item* up, void* act, int m, void* stream,
                        const wqaa_epilogue* norm) {
  if (!gate || !up || !gate->desc || !up->desc || !valid_desc(gate->desc) || !valid_desc(up->desc)) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_gate_up: missing item or descriptor");
    return WQAA_ERR_BAD_DESC;
  }
  if (m == 0) return WQAA_OK;
  if (norm && (norm->struct_size != (int32_t)sizeof(wqaa_epilogue) || norm->flags != WQAA_EPI_RMSNORM_INPUT || !norm->norm_weight)) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_gate_up: the epilogue must be a whole descriptor with WQAA_EPI_RMSNORM_INPUT and its weight");
    return WQAA_ERR_BAD_DESC;
  }
  const wqaa_matmul_desc& d = *gate->desc;
  if (memcmp(gate->desc, up->desc, sizeof(wqaa_matmul_desc)) != 0 || gate->A != up->A) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_gate_up: gate and up must share one input and agree in their descriptors");
    return WQAA_ERR_BAD_DESC;
  }
  if (m < 0 || !gate->A || !gate->B || !up->B || !act || (d.with_scaling && (!gate->Scale || !up->Scale)) ||
      (d.zeros_mode != WQAA_Z_NONE && (!gate->Zeros || !up->Zeros)) || (d.with_bias && (!gate->Bias || !up->Bias))) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_gate_up: bad call (m=%d, a NULL operand the descriptor asks for)", m);
    return WQAA_ERR_BAD_DESC;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  StreamDeviceScope scope(s, gate->A);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  int st = gemvx_pair_launch(d, gate, up, act, m, s, norm);
  if (st == WQAA_OK) g_last_error = WQAA_OK;
  return st;
}

int wqaa_gate_up_plan(const wqaa_matmul_desc* desc, int m, int with_norm, wqaa_plan* plan) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  return gemvx_pair_plan(*desc, m, plan, with_norm != 0);
}

int wqaa_matmul_group_ex(const wqaa_group_item* items, const wqaa_epilogue* const* epilogues, int count, int m, void* stream) {
  return group_impl(items, epilogues, count, m, stream);
}

int wqaa_matmul_chain(const wqaa_chain_item* items, int count, int m, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const void* anchor = (items && count > 0) ? (items[0].A ? items[0].A : items[0].B) : nullptr;
  StreamDeviceScope scope(s, anchor);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  int st = chain_launch(items, count, m, s);
  if (st == WQAA_OK) g_last_error = WQAA_OK;
  return st;
}

int wqaa_chain_plan(const wqaa_chain_item* items, int count, int m, int* launches, wqaa_plan* plan) {
  return chain_plan(items, count, m <= 0 ? 1 : m, launches, plan);
}

int wqaa_tune(const wqaa_matmul_desc* desc, int m, void* stream) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  if (m <= 0) return WQAA_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  StreamDeviceScope scope(s);
  if (!device_info().ok) return WQAA_OK;                      // nothing to measure on
  const int saved = g_last_error;
  int st = WQAA_OK;
  if (dense_lib_eligible(*desc, m)) st = dense_lib_tune(*desc, m, s, nullptr);
  else st = gemm_two_pass_tune(*desc, m, s);
  g_plan_epoch.fetch_add(1, std::memory_order_relaxed);       // workspace sizes may have changed with the algorithm
  if (st == WQAA_OK) g_last_error = saved;
  return st;
}

int wqaa_dequantize(const wqaa_matmul_desc* desc, const void* B, const void* LUT, const void* Scale, const void* Zeros,
                    void* out, void* stream) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  if (!B || !out || (desc->with_scaling && !Scale) || (desc->zeros_mode != WQAA_Z_NONE && !Zeros) ||
      (desc->w_format == WQAA_W_NF && !LUT)) {
    set_error(WQAA_ERR_BAD_DESC, "dequantize: null operand the descriptor requires");
    return WQAA_ERR_BAD_DESC;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  StreamDeviceScope scope(s, B);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  int st = gemm_dequantize_launch(*desc, B, LUT, Scale, Zeros, out, s);
  if (st == WQAA_OK) g_last_error = WQAA_OK;
  return st;
}

int wqaa_act_quant_int8(const void* X, int64_t rows, int K, void* Q, float* S, void* stream) {
  StreamDeviceScope scope(reinterpret_cast<hipStream_t>(stream), X);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  return act_quant_launch(X, rows, K, Q, S, reinterpret_cast<hipStream_t>(stream));
}

int wqaa_select(const wqaa_matmul_desc* desc, int m, wqaa_plan* plan) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  if (plan) memset(plan, 0, sizeof(*plan));
  if (m <= 0) m = 1;
  g_plan_epoch.fetch_add(1, std::memory_order_relaxed);   // planning re-reads the tuning environment (ChoiceMemo)
  if (dense_lib_eligible(*desc, m)) return dense_lib_plan(*desc, m, plan);
  if (two_pass_planned(*desc, m)) return gemm_two_pass_plan(*desc, m, plan);
  bool use_gemm = false;
  dispatch(*desc, m, &use_gemm);
  return use_gemm ? gemm_plan(*desc, m, plan) : gemv_plan(*desc, m, plan);
}

// wqaa_pack_weight / wqaa_unpack_weight / wqaa_relayout_weight: csrc/wqaa_pack.hip (host only)

int wqaa_select_ex(const wqaa_matmul_desc* desc, int m, int epilogue_flags, wqaa_plan* plan) {
  if (!valid_desc(desc)) return WQAA_ERR_BAD_DESC;
  if (plan) memset(plan, 0, sizeof(*plan));
  if (m <= 0) m = 1;
  if (epilogue_flags < 0) return wqaa_select(desc, m, plan);
  g_plan_epoch.fetch_add(1, std::memory_order_relaxed);
  if (epilogue_flags & (WQAA_EPI_ADD_RESIDUAL | WQAA_EPI_RMSNORM_INPUT)) {
    // the float16 path's fused ops exist in the exact-product GEMV family only
    wqaa_epilogue e;
    memset(&e, 0, sizeof(e));
    e.flags = epilogue_flags;
    if (!gemvx_covers(*desc, m) || desc->out_dtype != WQAA_F16) {
      set_error(WQAA_ERR_UNSUPPORTED, "select_ex: residual add / RMSNorm input need float16 x 1/2/4-bit integer weights, float16 output, m <= 2");
      return WQAA_ERR_UNSUPPORTED;
    }
    return gemvx_plan(*desc, m, plan);
  }
  bool use_gemm = false;
  dispatch(*desc, m, &use_gemm);
  if ((epilogue_flags & WQAA_EPI_QUANTIZE_INPUT) && m <= 4) use_gemm = false;
  return use_gemm ? gemm_plan(*desc, m, plan, true) : gemv_plan(*desc, m, plan);
}

int wqaa_debug_decode(const void* packed_dev, int64_t nwords, int w_format, int bits, int layout,
                      int a_dtype, int strict_reference, const void* lut_dev, void* out_dev,
                      void* stream) {
  StreamDeviceScope scope(reinterpret_cast<hipStream_t>(stream), packed_dev);
  if (!device_info().ok) {
    set_error(WQAA_ERR_NO_DEVICE, "no HIP device visible");
    return WQAA_ERR_NO_DEVICE;
  }
  return debug_decode_launch(packed_dev, nwords, w_format, bits, layout, a_dtype, strict_reference,
                             lut_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
}

void wqaa_debug_row_blocks(int b, int grid, int n_blocks, int* out3) {
  const RowBlocks rb = xcd_row_blocks(b, grid, n_blocks);
  out3[0] = rb.first;
  out3[1] = rb.stride;
  out3[2] = rb.end;
}

void wqaa_debug_tile_of_block(int tiles_m, int tiles_n, int ksplit, int group_m, int block, int* out4) {
  gemm_debug_tile_of_block(tiles_m, tiles_n, ksplit, group_m, block, out4);
}

int wqaa_last_error(void) { return g_last_error; }
const char* wqaa_last_error_string(void) { return g_last_error_msg; }

}  // extern "C"
