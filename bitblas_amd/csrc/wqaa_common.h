// wqaa_common.h - shared host/device declarations of libwqaa_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstring>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/wqaa.h"

namespace wqaa {

// weight decode kinds: one static kernel family member per kind
enum DecodeKind : int {
  DK_INT4 = 0,   // uint4 / int4 (zero point folded)
  DK_INT2 = 1,
  DK_INT1 = 2,
  DK_INT8 = 3,   // uint8 / int8 weights
  DK_LUT4 = 4,   // nf4 (LUT from caller) / fp4_e2m1 (built-in table)
  DK_E4M3 = 5,
  DK_E5M2 = 6,
  DK_NATIVE = 7  // W stored in A_dtype
};

struct GemvArgs {
  const void* A;
  const void* B;
  const void* lut;
  const void* scale;
  const void* zeros;
  const void* bias;
  void* C;
  int m;            // activation rows in this call
  int N, K;
  int kg;           // groups per weight row (K / group_size)
  int g_log2;       // log2(group_size) or -1 when not a power of two
  int g;            // group size (elements)
  int gq_shift;     // lane chunks per group = g / E as a shift (-1: use gq_magic)
  uint32_t gq_magic;  // ceil(2^32 / (g / E)) for non power-of-two ratios
  int nc;           // 64-lane chunks per weight row
  int ncp;          // nc rounded up to the step depth (LDS activation slots, zero beyond K)
  int cpr;          // valid 16-byte lane chunks per weight row
  long row_bytes;   // bytes per weight row
  int has_bias;
  int out_dtype;    // wqaa_dtype
  int is_signed;    // WQAA_W_INT
  int fp4_table;    // DK_LUT4: 1 = built-in fp4 table, 0 = caller LUT
  int a_fmt;        // wqaa_dtype of A as stored (fp8 activations are widened while staging)
  int zq_row_bytes; // quantized zeros: bytes per group row (N*bits/8)
  const float* epi_row;   // fused caller epilogue (wqaa_matmul_ex): out = half(acc / epi_row[m] / epi_tensor)
  float epi_tensor;
  // K split across the waves of a workgroup (few-row shards): kw consecutive waves share a row group, wave part p takes
  // the lane-chunk steps [p * spp, (p + 1) * spp); the parts meet in LDS in a fixed order.  kw == 1: everything below unused.
  int kw;               // 1, or a divisor of the workgroup's wave count
  uint32_t kw_magic;    // ceil(2^16 / kw): wave / kw without an integer division
  int spp;              // steps per part = ceil(steps / kw)
  int n_rgb;            // row-group blocks: ceil(ceil(N / R) / (waves / kw)) - what the workgroups share out (xcd_row_blocks)
};

// One launch serves up to kGemvGroupMax independent operators of one tile configuration (wqaa_matmul_group): blockIdx.z
// names the operator, every operator gets gridDim.x x gridDim.y workgroups.  A single call is the group of one.
constexpr int kGemvGroupMax = 8;
struct GemvGroupArgs {
  GemvArgs p[kGemvGroupMax];
};

struct LaunchCfg {
  int grid_x, grid_y;
  int threads;
  int lds_bytes;
  hipStream_t stream;
  hipEvent_t start, stop;  // optional (nullptr): kernel begin/end timestamps
};

void set_error(int code, const char* fmt, ...);

// kernel families (each returns a wqaa_status)
int gemv_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan);
int gemv_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi = nullptr);
void gemv_init();

// exact-product GEMV members (strict_reference = 0, sub-byte integer weights x float16, M <= 2)
bool gemvx_eligible(const wqaa_matmul_desc& d, int m);
bool gemvx_covers(const wqaa_matmul_desc& d, int m);
// groups: `merged` = the members' descriptor with N = the sum of their rows (what selects the tile configuration)
bool gemvx_group_eligible(const wqaa_matmul_desc& merged, const wqaa_matmul_desc* const* descs, int count, int m, bool norm = false);
int gemvx_group_plan(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, wqaa_plan* plan);
int gemvx_group_launch(const wqaa_matmul_desc& merged, const wqaa_group_item* items, int count, int m, hipStream_t stream,
                       const wqaa_epilogue* norm = nullptr);
bool gemv_group_eligible(const wqaa_matmul_desc& merged, const wqaa_matmul_desc* const* descs, int count, int m, bool with_epilogue = false, bool quant_in = false);
int gemv_group_plan(const wqaa_matmul_desc& merged, const int* Ns, int count, int m, wqaa_plan* plan);
int gemv_group_launch(const wqaa_matmul_desc& merged, const wqaa_group_item* items, int count, int m, hipStream_t stream,
                      const wqaa_epilogue* const* epis = nullptr);
int gemvx_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan);
int gemvx_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* Scale, const void* Zeros,
                 const void* Bias, void* C, int m, hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi = nullptr);
int gemvx_pair_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan, bool norm);
int gemvx_pair_launch(const wqaa_matmul_desc& d, const wqaa_group_item* gate, const wqaa_group_item* up, void* act, int m,
                      hipStream_t stream, const wqaa_epilogue* norm);
struct GemvxArgs;
// chains of dependent operators, run as the launches they stand for (wqaa_chain.hip)
int chain_launch(const wqaa_chain_item* items, int count, int m, hipStream_t stream);
int chain_plan(const wqaa_chain_item* items, int count, int m, int* launches, wqaa_plan* plan);
void gemvx_init();

void gemm_debug_tile_of_block(int tiles_m, int tiles_n, int ksplit, int group_m, int block, int* out4);
int gemm_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan, bool fused_epilogue = false);
bool pool_workspace_ready(hipStream_t stream, size_t bytes);
int gemm_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT,
                const void* Scale, const void* Zeros, const void* Bias, void* C, int m,
                hipStream_t stream, hipEvent_t start, hipEvent_t stop, const wqaa_epilogue* epi = nullptr,
                const wqaa_call_opts* opts = nullptr);
size_t gemm_workspace_bytes(const wqaa_matmul_desc& d, int m);
// two-pass member (large M): B_decode to a scratch by wq_dequant_kernel, then the plain GEMM through the vendor library
bool gemm_two_pass_eligible(const wqaa_matmul_desc& d, int m);
int gemm_two_pass_tune(const wqaa_matmul_desc& d, int m, hipStream_t stream);
int gemm_two_pass_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan);
size_t gemm_two_pass_workspace_bytes(const wqaa_matmul_desc& d, int m);
int gemm_two_pass_launch(const wqaa_matmul_desc& d, const void* A, const void* B, const void* LUT, const void* Scale, const void* Zeros,
                         void* C, int m, hipStream_t stream, const wqaa_call_opts* opts);
int gemm_dequantize_launch(const wqaa_matmul_desc& d, const void* B, const void* LUT, const void* Scale, const void* Zeros, void* out,
                           hipStream_t stream);
int act_quant_launch(const void* X, int64_t rows, int K, void* Q, float* S, hipStream_t stream);
void gemm_init();

// the library's scratch pool: one slab per (device, stream), retired - never freed - when it has to grow (wqaa_gemm.hip)
void* pool_workspace(hipStream_t stream, size_t bytes);

// plain dense GEMMs (W_dtype == A_dtype, a float type, no scale / zeros / bias, M >= 16) through hipBLASLt (wqaa_dense_lib.hip)
bool dense_lib_eligible(const wqaa_matmul_desc& d, int m, bool second_pass = false);
int dense_lib_plan(const wqaa_matmul_desc& d, int m, wqaa_plan* plan);
size_t dense_lib_workspace_bytes(const wqaa_matmul_desc& d, int m);
int dense_lib_tune(const wqaa_matmul_desc& d, int m, hipStream_t stream, float* best_ms);
int dense_lib_launch(const wqaa_matmul_desc& d, const void* A, const void* B, void* C, int m, hipStream_t stream,
                     const wqaa_call_opts* opts);

int debug_decode_launch(const void* packed, int64_t nwords, int w_format, int bits, int layout,
                        int a_dtype, int strict, const void* lut, void* out, hipStream_t stream);

// Tile choices are memoised per thread: a selector reads tuning environment variables and walks its rules, ~1 us
// per call against a 4 us kernel (twice per wqaa_matmul: family dispatch + launch).  wqaa_select() - operator
// planning - bumps the epoch, so the tuning variables are PLAN-time switches: a change takes effect at the next
// wqaa_select() of any operator, not in the middle of a stream of calls.
extern std::atomic<unsigned> g_plan_epoch;

// Tuning / test aids.  TWO environment variables hold them all, each a comma-separated list of key[=value] tokens read at plan
// time (the choices are memoised per plan epoch: wqaa_select and the plan queries bump it):
//   WQAA_GEMV_TUNE   exact=0|1|2  kw=N  grid=N  group_grid=N  areg=0|1  chunk=0
//   WQAA_GEMM_TUNE   ksplit=N  mid=0|1|2  pp_tile=0|128|256  pp_bn=128  pp_tail=0|1  pp8_wide=0|1  wide=0  ws_policy=BITS
//                    decode=0|1  decode_force=0|1  decode_persist=0  decode_long=0..4
// (README.md "Tuning variables" says what each key pins; the parity tests use them to put one member form next to another.)
// knob(var, key, &v): true and v = the token's value (1 for a bare key) when `var` holds the key.
inline bool knob(const char* var, const char* key, int* value) {
  const char* s = getenv(var);
  if (!s) return false;
  const size_t kl = strlen(key);
  while (*s) {
    const char* e = strchr(s, ',');
    const size_t n = e ? (size_t)(e - s) : strlen(s);
    if (n >= kl && strncmp(s, key, kl) == 0 && (n == kl || s[kl] == '=')) {
      *value = n == kl ? 1 : atoi(s + kl + 1);
      return true;
    }
    if (!e) break;
    s = e + 1;
  }
  return false;
}
inline bool gemv_knob(const char* key, int* value) { return knob("WQAA_GEMV_TUNE", key, value); }
inline bool gemm_knob(const char* key, int* value) { return knob("WQAA_GEMM_TUNE", key, value); }
inline bool gemm_knob_set(const char* key) { int v; return gemm_knob(key, &v); }
template <class Choice, int WAYS = 16>
struct ChoiceMemo {
  struct Entry {
    wqaa_matmul_desc d;
    int m, q, valid;
    unsigned epoch;
    Choice c;
  };
  Entry e[WAYS] = {};
  static unsigned slot(const wqaa_matmul_desc& d, int m, int q) {
    return ((unsigned)d.N * 2654435761u ^ (unsigned)d.K * 40503u ^ (unsigned)m * 97u ^ (unsigned)q ^
            ((unsigned)d.w_bits << 3) ^ ((unsigned)d.zeros_mode << 7) ^ ((unsigned)d.a_dtype << 11)) % WAYS;
  }
  const Choice* find(const wqaa_matmul_desc& d, int m, int q) const {
    const Entry& en = e[slot(d, m, q)];
    if (en.valid && en.m == m && en.q == q && en.epoch == g_plan_epoch.load(std::memory_order_relaxed) &&
        memcmp(&en.d, &d, sizeof(d)) == 0)
      return &en.c;
    return nullptr;
  }
  void put(const wqaa_matmul_desc& d, int m, int q, const Choice& c) {
    Entry& en = e[slot(d, m, q)];
    en.d = d; en.m = m; en.q = q; en.c = c;
    en.epoch = g_plan_epoch.load(std::memory_order_relaxed);
    en.valid = 1;
  }
};

struct DeviceInfo {
  int ok;
  int cus;
  int lds_per_block;
  char arch[64];
};
const DeviceInfo& device_info();   // of the CURRENT device, lazily initialised (kernel attributes included)
int current_device();

template <typename K>
inline hipError_t launch_kernel(K kernel, const LaunchCfg& cfg, void* args_struct) {
  void* params[] = {args_struct};
  dim3 grid(cfg.grid_x, cfg.grid_y, 1), block(cfg.threads, 1, 1);
  if (cfg.start != nullptr || cfg.stop != nullptr) {
    return hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, params,
                              cfg.lds_bytes, cfg.stream, cfg.start, cfg.stop, 0);
  }
  return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, params, cfg.lds_bytes,
                         cfg.stream);
}

// kernel-name fragments (general_matmul/__init__.py:240-318 spelling: f16, i4, u4, ...)
inline const char* short_dtype(int dt) {
  switch (dt) {
    case WQAA_F16: return "f16"; case WQAA_BF16: return "bf16"; case WQAA_F32: return "f32";
    case WQAA_I8: return "i8"; case WQAA_I32: return "i32"; case WQAA_E4M3: return "e4m3"; case WQAA_E5M2: return "e5m2";
    case WQAA_I4: return "i4";
  }
  return "x";
}
inline void short_wdtype(const wqaa_matmul_desc& d, char* buf, size_t n) {
  switch (d.w_format) {
    case WQAA_W_UINT: snprintf(buf, n, "u%d", d.w_bits); break;
    case WQAA_W_INT: snprintf(buf, n, "i%d", d.w_bits); break;
    case WQAA_W_NF: snprintf(buf, n, "nf%d", d.w_bits); break;
    case WQAA_W_FP4: snprintf(buf, n, "fp4_e2m1"); break;
    case WQAA_W_E4M3: snprintf(buf, n, "e4m3"); break;
    case WQAA_W_E5M2: snprintf(buf, n, "e5m2"); break;
    default: snprintf(buf, n, "%s", short_dtype(d.a_dtype));
  }
}


// (round 2's A/B aid WQAA_GEMV_UNCAP - GEMV grids not capped at the workgroups the chip holds at once - was settled by
// profiles/r02_ab_cap.txt and is gone: round 5's prune)
inline bool gemv_uncapped() { return false; }

inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace wqaa
