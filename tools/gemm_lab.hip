// gemm_lab.hip - A/B harness for the ping-pong MFMA member (csrc/wqaa_gemm_pp_kernel.h): every variant below is built into
// this one binary, checked against the library's shipped member through the C ABI (same operands) and timed in
// interleaved rounds with events on the launch stream.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bitblas_amd/csrc tools/gemm_lab.hip \
//         -L bitblas_amd -lwqaa_hip -Wl,-rpath,'$ORIGIN/../bitblas_amd' -o tools/gemm_lab
//   tools/gemm_lab [M N K] [--kind u4|i2] [--rounds R] [--iters I] [--only name]
#include "mm_lab_kernel.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

using namespace wqaa;

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)


// every variant once: (name, policy arguments)
#define LAB_F16_VARIANTS(X)                                                                 \
  X("pp", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, 0)                                      \
  X("pp_nozint", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ZINT_OFF)                    \
  X("pp_metaslow", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_META_SLOW)                 \
  X("pp_rw64", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_RW64)                          \
  X("pp_metaonce", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_METAONCE)             \
  X("pp_prio", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_PRIO)                          \
  X("pp_scaleonly", DK_INT4, LAYOUT_LOP3, AT_F16, MD_S, 0, 3, 0)                             \
  X("pp_nometa", DK_INT4, LAYOUT_LOP3, AT_F16, MD_NONE, 0, 3, 0)                             \
  X("pp_nometa_r4", DK_INT4, LAYOUT_LOP3, AT_F16, MD_NONE, 0, 4, 0)                          \
  X("pp_trace", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_TRACE)                        \
  X("abl_nodma", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_NODMA)                   \
  X("abl_noread", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_NOREAD)                 \
  X("abl_nodec", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_NODEC)                   \
  X("abl_all", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC)      \
  X("abl_all_long", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC | PPO_ABL_LONGSEG) \
  X("pp_long", DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 3, PPO_ABL_LONGSEG)
#define LAB_I8_VARIANTS(X)                                                                  \
  X("pp_i2", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, 0)                                  \
  X("pp_i2_r4", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 4, 0)                               \
  X("pp_i2_prio", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_PRIO)                      \
  X("pp_i2_ntaw", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, 0)           \
  X("abl_i2_nodma", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_ABL_NODMA)               \
  X("abl_i2_nodec", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_ABL_NODEC)               \
  X("abl_i2_all", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC)  \
  X("abl_i2_all_long", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC | PPO_ABL_LONGSEG) \
  X("pp_i2_long", DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 3, PPO_ABL_LONGSEG)
#define LAB_PUSH(name, ...) vs.push_back(mk<PPPolicy<__VA_ARGS__>>(name));


struct Variant {
  const char* name;
  gemm_fn fn;
  int lds;
  int bm = 256;
};

template <class P>
static Variant mk(const char* name) {
  gemm_fn fn = wq_gemm_pp_kernel<P>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return Variant{name, fn, P::LDS_BYTES, P::BM};
}

template <class P>
static Variant mk8(const char* name) {
  gemm_fn fn = wq_gemm_pp8_kernel<P>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return Variant{name, fn, P::LDS_BYTES};
}

template <class P>
static Variant mkmm(const char* name) {
  gemm_fn fn = wq_gemm_mm_kernel<P>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return Variant{name, fn, P::LDS_BYTES};
}

static uint32_t rng_state = 12345u;
static inline uint32_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return rng_state;
}
static inline float frand() { return (rng() >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
  int M = 4096, N = 4096, K = 4096, rounds = 5, iters = 20;
  std::string kind = "u4", only;
  int nmf_arg = 0;
  std::vector<int> pos;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s == "--kind" && i + 1 < argc) kind = argv[++i];
    else if (s == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (s == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (s == "--only" && i + 1 < argc) only = argv[++i];
    else if (s == "--nmf" && i + 1 < argc) nmf_arg = atoi(argv[++i]);
    else pos.push_back(atoi(argv[i]));
  }
  if (pos.size() >= 3) { M = pos[0]; N = pos[1]; K = pos[2]; }
  const bool i2 = kind == "i2", f8 = kind == "f8";
  const int G = 128;
  if (f8) setenv("WQAA_DENSE_LIB", "0", 1);     // the reference is this library's own lockstep member, not the vendor's
  init();

  // ---- operands (bench.py's: A = rand - 0.5, random codes, Scale = rand * 0.02, Zeros = 2^(bits-1)) ----
  const int bits = i2 ? 2 : f8 ? 8 : 4;
  const size_t a_bytes = (size_t)M * K * ((i2 || f8) ? 1 : 2), w_bytes = (size_t)N * K * bits / 8, c_bytes = (size_t)M * N * (i2 ? 4 : 2);
  const size_t sz_meta = (size_t)N * (K / G) * 2;
  std::vector<uint8_t> hA(a_bytes), hW(w_bytes);
  std::vector<_Float16> hS(N * (K / G)), hZ(N * (K / G));
  if (i2) {
    for (auto& b : hA) b = (uint8_t)(rng() & 0xFF);
  } else if (f8) {
    for (auto& b : hA) b = (uint8_t)(rng() & 0xBF);          // e4m3, |x| < 2, no NaN
  } else {
    _Float16* p = reinterpret_cast<_Float16*>(hA.data());
    for (size_t i = 0; i < (size_t)M * K; ++i) p[i] = (_Float16)(frand() - 0.5f);
  }
  for (auto& s : hS) s = (_Float16)(frand() * 0.02f);
  for (auto& z : hZ) z = (_Float16)8.0f;
  const int NSETS = 4;
  void *dA, *dC0, *dC1, *dS, *dZ, *dW[NSETS];
  CK(hipMalloc(&dA, a_bytes));
  CK(hipMalloc(&dC0, c_bytes));
  CK(hipMalloc(&dC1, c_bytes));
  CK(hipMalloc(&dS, sz_meta));
  CK(hipMalloc(&dZ, sz_meta));
  CK(hipMemcpy(dA, hA.data(), a_bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(dS, hS.data(), sz_meta, hipMemcpyHostToDevice));
  CK(hipMemcpy(dZ, hZ.data(), sz_meta, hipMemcpyHostToDevice));
  for (int s = 0; s < NSETS; ++s) {
    for (auto& b : hW) b = (uint8_t)(rng() & (f8 ? 0xBF : 0xFF));
    CK(hipMalloc(&dW[s], w_bytes));
    CK(hipMemcpy(dW[s], hW.data(), w_bytes, hipMemcpyHostToDevice));
  }

  // ---- the shipped member through the C ABI ----
  wqaa_matmul_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.N = N; d.K = K;
  d.a_dtype = i2 ? WQAA_I8 : f8 ? WQAA_E4M3 : WQAA_F16;
  d.w_format = i2 ? WQAA_W_INT : f8 ? WQAA_W_NATIVE : WQAA_W_UINT;
  d.w_bits = bits;
  d.out_dtype = i2 ? WQAA_I32 : WQAA_F16;
  d.group_size = (i2 || f8) ? -1 : G;
  d.with_scaling = (i2 || f8) ? 0 : 1;
  d.zeros_mode = (i2 || f8) ? WQAA_Z_NONE : WQAA_Z_ORIGINAL;
  d.w_layout = f8 ? WQAA_LAYOUT_PLAIN : WQAA_LAYOUT_LOP3;
  d.strict_reference = 1;
  wqaa_plan plan;
  if (wqaa_select(&d, M, &plan) != WQAA_OK) { fprintf(stderr, "select: %s\n", wqaa_last_error_string()); return 2; }
  printf("reference member: %s\n", plan.name);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto run_ref = [&](int set, void* C) {
    if (wqaa_matmul(&d, dA, dW[set], nullptr, (i2 || f8) ? nullptr : dS, (i2 || f8) ? nullptr : dZ, nullptr, C, M, st) != WQAA_OK) {
      fprintf(stderr, "matmul: %s\n", wqaa_last_error_string());
      exit(2);
    }
  };

  // ---- mid-M streaming member (--kind mm): every split count against the shipped member, partial sums reduced on the host ----
  if (kind == "mm") {
    const int nmf = nmf_arg ? nmf_arg : M <= 32 ? 2 : M <= 64 ? 4 : 8;      // tile rows / 16
    Variant v = nmf == 2 ? mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 1, 2>>("mm32") : nmf == 4 ? mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 2>>("mm64")
                                                                                             : mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4>>("mm128");
    run_ref(0, dC0);
    CK(hipStreamSynchronize(st));
    std::vector<_Float16> h0((size_t)M * N);
    CK(hipMemcpy(h0.data(), dC0, c_bytes, hipMemcpyDeviceToHost));
    double sr = 0;
    for (auto x : h0) sr += (double)(float)x * (double)(float)x;
    const double rms = std::sqrt(sr / ((double)M * N));
    float* dWS;
    CK(hipMalloc(&dWS, (size_t)16 * M * N * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int ks : {1, 2, 4, 8, 16}) {
      if (K / 256 < ks) continue;
      GemmArgs b;
      memset(&b, 0, sizeof(b));
      b.A = dA; b.B = dW[0]; b.scale = dS; b.zeros = dZ; b.C = dC1; b.M = M; b.N = N; b.K = K; b.kg = K / G; b.gq_shift = 0;
      b.row_bytes = (long)K / 2; b.out_dtype = WQAA_F16; b.tiles_m = (M + 16 * nmf - 1) / (16 * nmf); b.tiles_n = (N + 127) / 128;
      b.ksplit = ks; b.ws = dWS;
      const int grid = b.tiles_m * b.tiles_n * ks;
      void* params[] = {&b};
      CK(hipMemset(dC1, 0xFF, c_bytes));
      CK(hipMemset(dWS, 0xFF, (size_t)16 * M * N * 4));
      CK(hipLaunchKernel(reinterpret_cast<const void*>(v.fn), dim3(grid), dim3(512), params, v.lds, st));
      CK(hipStreamSynchronize(st));
      std::vector<float> got((size_t)M * N, 0.f);
      if (ks == 1) {
        std::vector<_Float16> h1((size_t)M * N);
        CK(hipMemcpy(h1.data(), dC1, c_bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) got[i] = (float)h1[i];
      } else {
        std::vector<float> part((size_t)ks * M * N);
        CK(hipMemcpy(part.data(), dWS, part.size() * 4, hipMemcpyDeviceToHost));
        for (int sidx = 0; sidx < ks; ++sidx)
          for (size_t i = 0; i < got.size(); ++i) got[i] += part[(size_t)sidx * M * N + i];
        for (auto& x : got) x = (float)(_Float16)x;
      }
      size_t nbad = 0, nan = 0;
      double worst = 0;
      for (size_t i = 0; i < got.size(); ++i) {
        const double p = (double)(float)h0[i], q = got[i];
        if (!(q == q)) { ++nan; continue; }
        worst = std::max(worst, std::fabs(p - q));
        if (std::fabs(p - q) > 1e-3 * std::fabs(p) + 1.5e-3 * rms) ++nbad;
      }
      std::vector<double> us;
      for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it) {
          b.B = dW[it % NSETS];
          CK(hipLaunchKernel(reinterpret_cast<const void*>(v.fn), dim3(grid), dim3(512), params, v.lds, st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) us.push_back(ms * 1e3 / iters);
      }
      std::sort(us.begin(), us.end());
      printf("%s ksplit %2d grid %4d : max|diff| %.3e outside %zu nan %zu%s   %8.2f us (main kernel only)\n", v.name, ks, grid, worst, nbad, nan,
             (nbad || nan) ? "  <-- FAIL" : "", us[us.size() / 2]);
    }
    // ablations of the 128-row tile at ksplit 1 (timing only)
    if (nmf == 8) {
      struct { const char* name; Variant v; } abl[] = {
          {"full", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4>>("a")},
          {"no dma in loop", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODMA>>("a")},
          {"no A reads", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NOREAD>>("a")},
          {"no decode", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODEC>>("a")},
          {"no barrier", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NOBAR>>("a")},
          {"no mfma", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NOMFMA>>("a")},
          {"no dma, no reads", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODMA | PPO_ABL_NOREAD>>("a")},
          {"no dma, reads, decode", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC>>("a")},
          {"barrier + mfma only", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC>>("a")},
          {"mfma only", mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC | PPO_ABL_NOBAR>>("a")},
      };
      for (int aks : {1, 8})
      for (auto& ab : abl) {
        GemmArgs b;
        memset(&b, 0, sizeof(b));
        b.A = dA; b.B = dW[0]; b.scale = dS; b.zeros = dZ; b.C = dC1; b.M = M; b.N = N; b.K = K; b.kg = K / G; b.gq_shift = 0;
        b.row_bytes = (long)K / 2; b.out_dtype = WQAA_F16; b.tiles_m = (M + 127) / 128; b.tiles_n = (N + 127) / 128;
        b.ksplit = aks; b.ws = dWS;
        const int grid = b.tiles_m * b.tiles_n * aks;
        void* params[] = {&b};
        std::vector<double> us;
        for (int r = 0; r < rounds + 1; ++r) {
          CK(hipEventRecord(e0, st));
          for (int it = 0; it < iters; ++it) CK(hipLaunchKernel(reinterpret_cast<const void*>(ab.v.fn), dim3(grid), dim3(512), params, ab.v.lds, st));
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (r) us.push_back(ms * 1e3 / iters);
        }
        std::sort(us.begin(), us.end());
        printf("ablation ksplit %d: %-24s %8.2f us\n", aks, ab.name, us[us.size() / 2]);
      }
    }
    // in-kernel time line (s_memrealtime, 100 MHz): every wave's stamps relative to the earliest start of the launch
    for (int ks : {1, 4, 8}) {
      Variant vt = nmf == 2 ? mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 1, 2, PPO_TRACE>>("mm32t")
                            : nmf == 4 ? mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 2, PPO_TRACE>>("mm64t") : mkmm<MMPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, 2, 4, PPO_TRACE>>("mm128t");
      GemmArgs b;
      memset(&b, 0, sizeof(b));
      b.A = dA; b.B = dW[0]; b.scale = dS; b.zeros = dZ; b.C = dC1; b.M = M; b.N = N; b.K = K; b.kg = K / G; b.gq_shift = 0;
      b.row_bytes = (long)K / 2; b.out_dtype = WQAA_F16; b.tiles_m = (M + 16 * nmf - 1) / (16 * nmf); b.tiles_n = (N + 127) / 128;
      b.ksplit = ks; b.ws = dWS;
      const int grid = b.tiles_m * b.tiles_n * ks;
      unsigned long long* dT;
      CK(hipMalloc(&dT, (size_t)grid * 64 * 8));
      CK(hipMemset(dT, 0, (size_t)grid * 64 * 8));
      b.lut = dT;
      void* params[] = {&b};
      for (int rep = 0; rep < 3; ++rep) {
        b.B = dW[rep % NSETS];
        CK(hipLaunchKernel(reinterpret_cast<const void*>(vt.fn), dim3(grid), dim3(512), params, vt.lds, st));
      }
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hT((size_t)grid * 64);
      CK(hipMemcpy(hT.data(), dT, hT.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (size_t i = 0; i < hT.size(); i += 8) if (hT[i]) t0 = std::min(t0, hT[i]);
      const char* nm[7] = {"start", "setup done", "dma issued", "first data + barrier", "loop done", "dma drained", "stores done"};
      printf("time line ksplit %d (us after the first wave's start; median [min .. max] over waves):\n", ks);
      for (int j = 0; j < 7; ++j) {
        std::vector<double> v;
        for (size_t i = 0; i < hT.size(); i += 8) if (hT[i + j]) v.push_back((double)(hT[i + j] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-22s %7.2f [%7.2f .. %7.2f]\n", nm[j], v[v.size() / 2], v.front(), v.back());
      }
      CK(hipFree(dT));
    }
    std::vector<double> us;
    for (int r = 0; r < rounds + 1; ++r) {
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) run_ref(it % NSETS, dC0);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) us.push_back(ms * 1e3 / iters);
    }
    std::sort(us.begin(), us.end());
    printf("shipped member (all its launches) : %8.2f us\n", us[us.size() / 2]);
    return 0;
  }

  // ---- variants ----
  std::vector<Variant> vs;
  if (f8) {
    vs.push_back(mk8<PP8Policy<0, 0>>("pp8_e4m3"));
  } else if (!i2) {
    LAB_F16_VARIANTS(LAB_PUSH)
    vs.push_back(mk<PPPolicy<DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 5, 0, 128>>("pp128"));
    vs.push_back(mk<PPPolicy<DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 5, PPO_ABL_NODMA | PPO_ABL_NOREAD | PPO_ABL_NODEC, 128>>("abl128_all"));
  } else {
    LAB_I8_VARIANTS(LAB_PUSH)
    vs.push_back(mk<PPPolicy<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0, 5, 0, 128>>("pp128_i2"));
  }
  if (!only.empty()) {             // --only a,b,c
    const std::string list = "," + only + ",";
    vs.erase(std::remove_if(vs.begin(), vs.end(), [&](const Variant& v) { return list.find("," + std::string(v.name) + ",") == std::string::npos; }), vs.end());
  }

  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = dA; a.scale = dS; a.zeros = dZ; a.C = dC1;
  a.M = M; a.N = N; a.K = K;
  a.kg = (i2 || f8) ? 1 : K / G;
  a.gq_shift = (i2 || f8) ? 0 : 0;                 // k-bodies per group, as a shift (g = 128: one body per group)
  a.row_bytes = (long)K * bits / 8;
  a.out_dtype = i2 ? WQAA_I32 : WQAA_F16;
  a.is_signed = i2 ? 1 : 0;
  a.tiles_m = (M + 255) / 256;
  a.tiles_n = (N + 255) / 256;
  a.group_m = a.tiles_m >= 4 ? 4 : 1;
  a.ksplit = 1;
  auto run_var = [&](const Variant& v, int set, void* C) {
    GemmArgs b = a;
    b.B = dW[set];
    b.C = C;
    b.tiles_m = (M + v.bm - 1) / v.bm;
    b.group_m = b.tiles_m >= 4 ? 4 : 1;
    void* params[] = {&b};
    CK(hipLaunchKernel(reinterpret_cast<const void*>(v.fn), dim3(b.tiles_m * b.tiles_n), dim3(512), params, v.lds, st));
  };

  // ---- parity against the shipped member ----
  run_ref(0, dC0);
  CK(hipStreamSynchronize(st));
  std::vector<uint8_t> h0(c_bytes), h1(c_bytes);
  CK(hipMemcpy(h0.data(), dC0, c_bytes, hipMemcpyDeviceToHost));
  for (const Variant& v : vs) {
    if (strstr(v.name, "scaleonly") || strstr(v.name, "nometa") || strstr(v.name, "abl_") || false) continue;   // different arithmetic: timing only
    CK(hipMemset(dC1, 0xFF, c_bytes));
    run_var(v, 0, dC1);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h1.data(), dC1, c_bytes, hipMemcpyDeviceToHost));
    if (i2) {
      const int* x = reinterpret_cast<const int*>(h0.data());
      const int* y = reinterpret_cast<const int*>(h1.data());
      size_t bad = 0;
      for (size_t i = 0; i < (size_t)M * N; ++i) bad += x[i] != y[i];
      printf("parity %-18s : %zu of %zu int32 differ%s\n", v.name, bad, (size_t)M * N, bad ? "  <-- FAIL" : "  (bit exact)");
    } else {
      const _Float16* x = reinterpret_cast<const _Float16*>(h0.data());
      const _Float16* y = reinterpret_cast<const _Float16*>(h1.data());
      double se = 0, sr = 0, worst = 0;
      size_t nbad = 0, nan = 0;
      for (size_t i = 0; i < (size_t)M * N; ++i) {
        const double p = (double)(float)x[i], q = (double)(float)y[i];
        if (!(q == q)) { ++nan; continue; }
        se += (p - q) * (p - q);
        sr += p * p;
        worst = std::max(worst, std::fabs(p - q));
      }
      const double rms = std::sqrt(sr / ((double)M * N));
      for (size_t i = 0; i < (size_t)M * N; ++i) {
        const double p = (double)(float)x[i], q = (double)(float)y[i];
        if (std::fabs(p - q) > 1e-3 * std::fabs(p) + 1.5e-3 * rms) ++nbad;
      }
      printf("parity %-18s : rms(ref) %.4f  rms(diff) %.3e  max|diff| %.3e  outside(1e-3 rel + 1.5e-3 rms) %zu  nan %zu%s\n", v.name, rms,
             std::sqrt(se / ((double)M * N)), worst, nbad, nan, (nbad || nan) ? "  <-- FAIL" : "");
    }
  }

  // ---- phase trace (variant *_trace): medians over all waves of a group ----
  for (const Variant& v : vs) {
    if (!strstr(v.name, "trace")) continue;
    if (i2) continue;
    const int nblk = a.tiles_m * a.tiles_n;
    unsigned long long* dT;
    CK(hipMalloc(&dT, (size_t)nblk * 8 * 20 * 8));
    CK(hipMemset(dT, 0, (size_t)nblk * 8 * 20 * 8));
    GemmArgs b = a;
    b.B = dW[0];
    b.C = dC1;
    b.lut = dT;
    void* params[] = {&b};
    for (int rep = 0; rep < 3; ++rep) CK(hipLaunchKernel(reinterpret_cast<const void*>(v.fn), dim3(nblk), dim3(512), params, v.lds, st));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hT((size_t)nblk * 8 * 20);
    CK(hipMemcpy(hT.data(), dT, hT.size() * 8, hipMemcpyDeviceToHost));
    const char* seg[4] = {"L issue (reads+dma)", "L wait (lgkm/vm)", "barrier L->C + wait", "C (8 mfma + decode) + barrier"};
    for (int g = 0; g < 2; ++g) {
      printf("trace %s, waves %d-%d, k-tile 16, median clocks [p10 .. p90]:\n", v.name, g * 4, g * 4 + 3);
      for (int j = 0; j < 4; ++j)
        for (int sgm = 0; sgm < 4; ++sgm) {
          if (j == 3 && sgm == 3) continue;
          std::vector<long> dts;
          for (int blk = 0; blk < nblk; ++blk)
            for (int w = g * 4; w < g * 4 + 4; ++w) {
              const unsigned long long* t = &hT[((size_t)blk * 8 + w) * 20];
              const int i0 = j * 4 + sgm, i1 = i0 + 1;
              if (t[i0] && t[i1]) dts.push_back((long)(t[i1] - t[i0]));
            }
          if (dts.empty()) continue;
          std::sort(dts.begin(), dts.end());
          printf("  phase %d  %-32s %6ld  [%ld .. %ld]\n", j, seg[sgm], dts[dts.size() / 2], dts[dts.size() / 10], dts[dts.size() * 9 / 10]);
        }
    }
    {
      std::vector<double> ghz;
      for (int blk = 0; blk < nblk; ++blk) {
        const unsigned long long* t = &hT[((size_t)blk * 8) * 20];
        if (t[19] > t[17]) ghz.push_back((double)(t[18] - t[16]) / ((double)(t[19] - t[17]) * 10.0));   // realtime: 100 MHz
      }
      std::sort(ghz.begin(), ghz.end());
      const unsigned long long* t0 = &hT[0];
      printf("  loop span of block 0 wave 0: %llu clocks, %.2f us (100 MHz realtime)  -> median %.3f GHz over blocks\n", t0[18] - t0[16],
             (double)(t0[19] - t0[17]) / 100.0, ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    }
    CK(hipFree(dT));
  }

  // ---- timing: interleaved rounds ----
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * M * N * K;
  std::vector<std::vector<double>> us(vs.size() + 1);
  for (int r = 0; r < rounds + 1; ++r) {       // round 0 = warm-up
    for (size_t vi = 0; vi <= vs.size(); ++vi) {
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) {
        if (vi == vs.size()) run_ref(it % NSETS, dC0);
        else run_var(vs[vi], it % NSETS, dC1);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) us[vi].push_back(ms * 1e3 / iters);
    }
  }
  for (size_t vi = 0; vi <= vs.size(); ++vi) {
    auto& v = us[vi];
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2], mn = v[0];
    printf("time   %-18s : median %8.2f us  min %8.2f us  -> %7.1f T%s/s (median)  %.3f of peak\n", vi == vs.size() ? "shipped" : vs[vi].name, med,
           mn, flop / med / 1e6, i2 ? "OP" : "FLOP", flop / med / 1e6 / (i2 ? 5000.0 : 2500.0));
  }
  return 0;
}
