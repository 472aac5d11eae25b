// gemv_lab.hip - lab harness for the exact-product GEMV (csrc/wqaa_gemvx_kernel.h): in-kernel time lines of the shipped
// geometry (s_memrealtime stamps of every wave, lab-only policy bit 64) next to the library's launch on the same box, same buffers;
// --rotate-mb: how many MB of weight sets the launches rotate over (cache-residency experiment).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bitblas_amd/csrc -I include tools/gemv_lab.hip \
//         -L bitblas_amd -lwqaa_hip -Wl,-rpath,'$ORIGIN/../bitblas_amd' -o tools/gemv_lab
//   tools/gemv_lab N K [--group G] [--iters I] [--rounds R] [--pro P]
// --pro 3: the time line of the member with the RMSNorm in front (GemvxPolicy PRO = 3) next to the plain one; --pro 2 / 4: the gate / up
// pair (two operators of N rows each, without / with the norm) - lab launches only, the library's launch stays the plain operator
// int4 (signed, LOP3 layout), fp16 activations, scale per group of 128, M = 1: the headline configuration.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "wqaa.h"
#include "wqaa_gemvx_kernel.h"

using namespace wqaa;

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rng() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
static inline float frand() { return (float)(rng() & 0xFFFFFF) / 16777216.f; }


int main(int argc, char** argv) {
  int N = 4096, K = 4096, G = 128, iters = 200, rounds = 7, count = 1, rotate_mb = 600, pro = 0;
  std::vector<int> pos;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s == "--group" && i + 1 < argc) G = atoi(argv[++i]);
    else if (s == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (s == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (s == "--rotate-mb" && i + 1 < argc) rotate_mb = atoi(argv[++i]);   // weight sets rotate over this many MB (600: HBM-cold; 120: memory-side cache; 0: one set)
    else if (s == "--count" && i + 1 < argc) count = atoi(argv[++i]);     // operators of a group launch (q/k/v: 3 x 4096)
    else if (s == "--pro" && i + 1 < argc) pro = atoi(argv[++i]);         // 2 pair, 3 norm, 4 norm + pair (time line of that member)
    else pos.push_back(atoi(argv[i]));
  }
  if (pos.size() >= 2) { N = pos[0]; K = pos[1]; }

  const size_t w_bytes = (size_t)N * K / 2, meta = (size_t)N * (K / G);
  const int NSETS = std::max(1, (int)(((unsigned long long)rotate_mb << 20) / (w_bytes * count)));
  std::vector<_Float16> hA(K), hS(meta);
  for (auto& x : hA) x = (_Float16)(frand() - 0.5f);
  for (auto& x : hS) x = (_Float16)(frand() * 0.02f);
  std::vector<uint8_t> hW(w_bytes);
  void *dA, *dS;
  std::vector<void*> dW(NSETS * count), dC(count);
  CK(hipMalloc(&dA, K * 2));
  CK(hipMalloc(&dS, meta * 2));
  CK(hipMemcpy(dA, hA.data(), K * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dS, hS.data(), meta * 2, hipMemcpyHostToDevice));
  for (auto& p : dW) {
    for (auto& b : hW) b = (uint8_t)(rng() & 0xFF);
    CK(hipMalloc(&p, w_bytes));
    CK(hipMemcpy(p, hW.data(), w_bytes, hipMemcpyHostToDevice));
  }
  for (auto& p : dC) CK(hipMalloc(&p, N * 2));
  void* dC2;
  CK(hipMalloc(&dC2, (size_t)N * 2 * count));

  wqaa_matmul_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.N = N; d.K = K;
  d.a_dtype = WQAA_F16; d.w_format = WQAA_W_INT; d.w_bits = 4; d.out_dtype = WQAA_F16;
  d.group_size = G; d.with_scaling = 1; d.zeros_mode = WQAA_Z_NONE; d.w_layout = WQAA_LAYOUT_LOP3;
  d.strict_reference = 0;
  wqaa_matmul_desc dm = d;
  dm.N = N * count;                                   // the merged operator chooses the tile configuration of a group
  wqaa_plan plan;
  if (wqaa_select(&dm, 1, &plan) != WQAA_OK) { fprintf(stderr, "select: %s\n", wqaa_last_error_string()); return 2; }
  printf("library member: %s  (threads %d, grid %d, lds %d, rows/wave %d, k split %d)\n", plan.name, plan.threads, plan.grid, plan.lds_bytes,
         plan.rows_per_wave, plan.split_k);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  auto time_it = [&](auto&& launch) {
    std::vector<double> us;
    for (int r = 0; r < rounds + 1; ++r) {
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) launch(it % NSETS);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) us.push_back(ms * 1e3 / iters);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
  };
  const double bytes = (double)count * (w_bytes + meta * 2) + K * 2 + (double)count * N * 2;

  // ---- the library's launch (single operator, or the group call) ----
  auto lib_launch = [&](int set) {
    int rc;
    if (count == 1) {
      rc = wqaa_matmul(&d, dA, dW[set], nullptr, dS, nullptr, nullptr, dC[0], 1, st);
    } else {
      wqaa_group_item items[8];
      for (int i = 0; i < count; ++i) items[i] = wqaa_group_item{&d, dA, dW[set * count + i], nullptr, dS, nullptr, nullptr, dC[i]};
      rc = wqaa_matmul_group(items, count, 1, st);
    }
    if (rc != WQAA_OK) { fprintf(stderr, "matmul: %s\n", wqaa_last_error_string()); exit(2); }
  };
  lib_launch(0);
  CK(hipStreamSynchronize(st));
  const double t_lib = time_it(lib_launch);
  printf("%-44s %8.2f us  %7.0f GB/s\n", "library launch", t_lib, bytes / t_lib * 1e-3);

  // ---- the same geometry through the lab's own fill: plain, time line ----
  const int R = plan.rows_per_wave, kw = plan.split_k, nw = plan.threads / 64;
  const int E = 32, cpr = K / E, nc = (cpr + 63) / 64, D = 2, nsteps = (nc + D - 1) / D;
  auto fill = [&](GemvxArgs& a, const void* B, void* C, const void* bias) {
    memset(&a, 0, sizeof(a));
    a.A = dA; a.B = B; a.scale = dS; a.zeros = nullptr; a.bias = bias; a.C = C;
    a.m = 1; a.N = N; a.K = K; a.kg = K / G;
    const int dq = G / E;
    a.gq_shift = ilog2_exact(dq);
    a.gq_magic = a.gq_shift >= 0 ? 0u : (uint32_t)(((1ull << 32) + dq - 1) / dq);
    a.nc = nc; a.cpr = cpr; a.nsteps = nsteps; a.kw = kw;
    a.row_bytes = (long)K / 2;
    a.has_bias = 0; a.out_dtype = WQAA_F16; a.zint = 8; a.flip = 0; a.zq_row_bytes = N / 2;
    const int slots = nw / kw;
    a.n_rgb = ((N + R - 1) / R + slots - 1) / slots;
    a.slots = slots;
    a.kw_magic = (65536u + (uint32_t)kw - 1u) / (uint32_t)kw;
  };
  const int gx = count == 1 ? plan.grid : plan.grid / count;
  gemvx_fn fn_plain = R == 2 ? wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2>> : wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 1, 2>>;
  gemvx_fn fn_trace = R == 2 ? wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 64>> : wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 1, 2, 64>>;
  // members with the caller's ops folded in (R = 2 only: what the q/k/v group and the gate / up pair run): same geometry, their
  // own kernels; a pair streams operator `set` and `set + 1` of the rotation as gate and up
  void* dNW = nullptr;
  if (pro >= 3) {
    std::vector<_Float16> hN(K);
    for (auto& x : hN) x = (_Float16)(0.75f + 0.5f * frand());
    CK(hipMalloc(&dNW, K * 2));
    CK(hipMemcpy(dNW, hN.data(), K * 2, hipMemcpyHostToDevice));
  }
  gemvx_fn fn_pro_plain = nullptr, fn_pro_trace = nullptr;
  if (pro == 2) { fn_pro_plain = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 0, false, 2>>; fn_pro_trace = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 64, false, 2>>; }
  if (pro == 3) { fn_pro_plain = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 0, false, 3>>; fn_pro_trace = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 64, false, 3>>; }
  if (pro == 4) { fn_pro_plain = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 0, false, 4>>; fn_pro_trace = wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 64, false, 4>>; }
  unsigned long long* dT;
  const size_t tr_words = (size_t)gx * count * nw * 8;
  CK(hipMalloc(&dT, tr_words * 8));
  auto lab_launch = [&](gemvx_fn fn, int set, const void* bias) {
    GemvxGroupArgs ga;
    for (int i = 0; i < count; ++i) fill(ga.p[i], dW[set * count + i], (char*)dC2 + (size_t)i * N * 2, bias);
    void* params[] = {&ga};
    CK(hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(gx, count, 1), dim3(nw * 64), params, plan.lds_bytes, st));
  };
  auto pro_launch = [&](gemvx_fn fn, int set, const void* bias) {
    GemvxGroupArgs ga;
    const bool pair = pro == 2 || pro == 4;
    for (int i = 0; i < (pair ? 2 : count); ++i) {
      fill(ga.p[i], dW[(set * count + i) % (int)dW.size()], (char*)dC2 + (size_t)(pair ? 0 : i) * N * 2, bias);
      ga.p[i].norm_weight = dNW;
      ga.p[i].norm_eps = 1e-5f;
      ga.p[i].norm_inv_k = 1.f / (float)K;
      if (pair) {               // a row group is ONE output element: N pairs
        const int slots = nw / kw;
        ga.p[i].n_rgb = (N + slots - 1) / slots;
      }
    }
    void* params[] = {&ga};
    CK(hipLaunchKernel(reinterpret_cast<const void*>(fn), dim3(pair ? 2 * gx : gx, pair ? 1 : count, 1), dim3(nw * 64), params, plan.lds_bytes, st));
  };
  if (pro >= 2 && R == 2 && !strstr(plan.name, "areg") && (size_t)count * NSETS >= 2) {
    const char* what = pro == 2 ? "gate / up pair" : pro == 3 ? "RMSNorm in front" : "RMSNorm + gate / up pair";
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn_pro_plain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn_pro_trace), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const double t_pro = time_it([&](int set) { pro_launch(fn_pro_plain, set, nullptr); });
    printf("%-44s %8.2f us  (%s; a pair streams two operators)\n", "member with the fused op, lab launch", t_pro, what);
    const size_t words = (size_t)gx * ((pro == 2 || pro == 4) ? 2 : count) * nw * 8;
    unsigned long long* dTp;
    CK(hipMalloc(&dTp, words * 8));
    CK(hipMemset(dTp, 0, words * 8));
    for (int rep = 0; rep < 4; ++rep) pro_launch(fn_pro_trace, rep % NSETS, dTp);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hT(words);
    CK(hipMemcpy(hT.data(), dTp, words * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < hT.size(); i += 8) if (hT[i]) t0 = std::min(t0, hT[i]);
    const char* nm[8] = {"wave start", "first weight loads issued", "activations staged + barrier", "first weights landed", "first position consumed",
                         "last position consumed", "stored", "norm: sum of squares known"};
    printf("time line of the member with the fused op (%s), last of 4 launches (us after the first wave's start; median [min .. p90 .. max]):\n", what);
    for (int j : {0, 1, 7, 2, 3, 4, 5, 6}) {
      std::vector<double> v;
      for (size_t i = 0; i < hT.size(); i += 8) if (hT[i + j] && hT[i]) v.push_back((double)(hT[i + j] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-30s %7.2f [%7.2f .. %7.2f .. %7.2f]   (%zu waves)\n", nm[j], v[v.size() / 2], v.front(), v[v.size() * 9 / 10], v.back(), v.size());
    }
  }
  if (!strstr(plan.name, "areg")) {
    lib_launch(0);
    lab_launch(fn_plain, 0, nullptr);
    CK(hipStreamSynchronize(st));
    std::vector<_Float16> h0(N), h1(N);
    CK(hipMemcpy(h0.data(), dC[0], N * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), dC2, N * 2, hipMemcpyDeviceToHost));
    printf("lab fill vs library: %s\n", memcmp(h0.data(), h1.data(), N * 2) == 0 ? "bit-identical" : "DIFFERENT");
    const double t_plain = time_it([&](int set) { lab_launch(fn_plain, set, nullptr); });
    printf("%-44s %8.2f us  %7.0f GB/s\n", "same kernel, lab launch", t_plain, bytes / t_plain * 1e-3);
    CK(hipMemset(dT, 0, tr_words * 8));
    for (int rep = 0; rep < 4; ++rep) lab_launch(fn_trace, rep % NSETS, dT);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hT(tr_words);
    CK(hipMemcpy(hT.data(), dT, tr_words * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < hT.size(); i += 8) if (hT[i]) t0 = std::min(t0, hT[i]);
    const char* nm[7] = {"wave start", "first weight loads issued", "activations staged + barrier", "first weights landed", "first position consumed",
                         "last position consumed", "stored"};
    printf("time line of the last of 4 launches (us after the first wave's start; median [min .. p90 .. max] over waves):\n");
    for (int j = 0; j < 7; ++j) {
      std::vector<double> v;
      for (size_t i = 0; i < hT.size(); i += 8) if (hT[i + j] && hT[i]) v.push_back((double)(hT[i + j] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("  %-30s %7.2f [%7.2f .. %7.2f .. %7.2f]   (%zu waves)\n", nm[j], v[v.size() / 2], v.front(), v[v.size() * 9 / 10], v.back(), v.size());
    }
  }

  return 0;
}
