// (round 6: the in-launch seam of round 5 was removed from the kernel; phases 5-7 below no longer exist and `spin_us` is ignored)
// mid_trace.hip - lab harness: per-wave phase timeline of the mid-M member (csrc/wqaa_gemm_mid_kernel.h) built with -DWQAA_TRACE.
// uint4, LOP3 layout, fp16, scale + zeros "original", g = 128; launches over rotating weight buffers; prints, for the last
// launch, the distribution over waves of the phase END times since the first wave's start (100 MHz s_memrealtime is taken at entry
// and exit; the phases in between are shader-clock stamps scaled to the wave's own entry..exit span):
//   0 entry | 1 loads + DMA issued | 2 landed + barrier | 3 multiply done | 4 k-halves met, published, stores acknowledged |
//   5 ticket back | 6 everybody here (or last) | 7 own portion reduced + stored | exit
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWQAA_TRACE -Ibitblas_amd/csrc -o tools/mid_trace tools/mid_trace.hip
//   tools/mid_trace M N K [spin_us]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "wqaa_gemm_mid_kernel.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace wqaa;

template <int MF, int NKH>
static void run(int M, int N, int K, int spin_us) {
  using P = MidPolicy<DK_INT4, LAYOUT_LOP3, MD_ZO, MF, NKH>;
  void (*fn)(const GemmArgs) = wq_gemm_mid_kernel<P>;
  const int lds = P::LDS_BYTES, nwaves = 8, g = 128, launches = 32;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t wbytes = (size_t)N * K / 2, sbytes = (size_t)N * (K / g) * 2;
  const int nbuf = (int)std::max<size_t>(2, (640ull << 20) / wbytes);
  std::vector<uint8_t> h(wbytes);
  srand(1);
  for (auto& b : h) b = (uint8_t)rand();
  std::vector<uint16_t> hs(sbytes / 2, 0x2200), hz(sbytes / 2, 0x4800), ha((size_t)M * K, 0x3400);
  std::vector<void*> W(nbuf), S(nbuf), Z(nbuf);
  for (int i = 0; i < nbuf; ++i) {
    CK(hipMalloc(&W[i], wbytes)); CK(hipMemcpy(W[i], h.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&S[i], sbytes)); CK(hipMemcpy(S[i], hs.data(), sbytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&Z[i], sbytes)); CK(hipMemcpy(Z[i], hz.data(), sbytes, hipMemcpyHostToDevice));
  }
  void *A, *C;
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&C, (size_t)M * N * 2));
  const int tiles_m = (M + P::BM - 1) / P::BM, tiles_n = (N + 127) / 128, tiles = tiles_m * tiles_n;
  const int grid = tiles * 8;
  const size_t trace_words = (size_t)grid * nwaves * 16;
  void* WS; CK(hipMalloc(&WS, (size_t)tiles * 64 * MF * 1024));

  unsigned long long* T;
  CK(hipMalloc(&T, trace_words * 8 * launches)); CK(hipMemset(T, 0, trace_words * 8 * launches));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int l = 0; l < launches; ++l) {
      GemmArgs a{};
      a.A = A; a.B = W[l % nbuf]; a.scale = S[l % nbuf]; a.zeros = Z[l % nbuf]; a.C = C;
      a.M = M; a.N = N; a.K = K; a.kg = K / g; a.gq_shift = 2; a.row_bytes = K / 2; a.out_dtype = 0; a.is_signed = 0;
      a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.nsteps = K / 128; a.group_m = 1; a.ksplit = 1; a.epi_tensor = 1.f;
      a.mg_ntiles = tile_magic((uint32_t)tiles_n);
      a.ws = WS;       // (round 6: the two-launch seam is the only one; the in-launch seam this harness also timed in round 5 is gone)
      a.lut = T + (size_t)l * trace_words;
      hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * nwaves), lds, st, a);
      if (spin_us < 0) hipLaunchKernelGGL(wq_mid_reduce_kernel<0>, dim3((tiles * 8 * MF + 3) / 4), dim3(256), 0, st, a, MF, tiles * 8 * MF);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M=%d N=%d K=%d mf=%d nkh=%d grid=%d spin=%dus: %.2f us per launch (eager, %d launches)\n", M, N, K, MF, NKH, grid, spin_us, ms * 1e3 / launches, launches);
  }
  std::vector<unsigned long long> t(trace_words);
  CK(hipMemcpy(t.data(), T + (size_t)(launches - 1) * trace_words, trace_words * 8, hipMemcpyDeviceToHost));
  unsigned long long first = ~0ull, last = 0;
  const size_t nw = (size_t)grid * nwaves;
  for (size_t w = 0; w < nw; ++w) { if (t[w * 16 + 8]) { first = std::min(first, t[w * 16 + 8]); last = std::max(last, t[w * 16 + 9]); } }
  printf("kernel span (first wave start -> last wave end): %.2f us\n", (last - first) * 0.01);
  const char* names[9] = {"entry", "issued", "landed+barrier", "multiplied", "met+published+acked", "ticket", "all here", "reduced+stored", "exit"};
  for (int ph = 0; ph < 9; ++ph) {
    std::vector<double> v;
    for (size_t w = 0; w < nw; ++w) {
      const unsigned long long* d = &t[w * 16];
      if (!d[8] || !d[0]) continue;
      const double start_us = (d[8] - first) * 0.01, span_us = (d[9] - d[8]) * 0.01;
      const unsigned long long c_end = std::max(std::max(d[7], d[4]), std::max(d[6], d[5]));    // last shader-clock stamp taken
      if (ph == 0) { v.push_back(start_us); continue; }
      if (ph == 8) { v.push_back(start_us + span_us); continue; }
      if (!d[ph]) continue;
      // shader clock -> us: scale by the wave's own (exit - entry) realtime over its last stamp (exit is a few hundred clocks after it)
      const double clk_per_us = c_end > d[0] ? (double)(c_end - d[0]) / std::max(span_us, 1e-3) : 2000.0;
      v.push_back(start_us + (double)(d[ph] - d[0]) / clk_per_us);
    }
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-22s  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us   (%zu waves)\n", names[ph], v.front(), v[v.size() / 10], v[v.size() / 2],
           v[v.size() * 9 / 10], v.back(), v.size());
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int spin = argc > 4 ? atoi(argv[4]) : 50;
  const int rows = M > 128 ? (M + (M + 127) / 128 - 1) / ((M + 127) / 128) : M;
  if (K == 4096) {
    if (rows > 64) run<8, 2>(M, N, K, spin);
    else if (rows > 32) run<4, 2>(M, N, K, spin);
    else run<2, 2>(M, N, K, spin);
  } else if (K == 2048) {
    if (rows > 64) run<8, 1>(M, N, K, spin);
    else run<4, 1>(M, N, K, spin);
  } else if (K == 8192) {
    if (rows > 32) run<4, 4>(M, N, K, spin);
    else run<2, 4>(M, N, K, spin);
  }
  return 0;
}
