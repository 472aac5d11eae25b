// decode_trace.hip - lab harness: per-wave phase timeline of the decode-LDS GEMM member (M <= 16).
// Includes the product kernel header with -DWQAA_TRACE, launches ONE member (uint4, LOP3 layout, fp16, scale + zeros
// "original", g = 128) over rotating weight buffers and prints, for the last launch, the distribution over waves of
//   start (since the first wave's start, 100 MHz clock), and the phase lengths in shader clocks:
//   entry -> loads issued -> k-step 0 done -> k-step 2 done -> MFMAs done -> barrier -> store
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWQAA_TRACE -Ibitblas_amd/csrc -o tools/decode_trace tools/decode_trace.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "wqaa_gemm_kernel.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace wqaa;

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  // 4th argument > 0: the pipelined 64-row member (8-byte metadata form, plan suffix xw) with that split-K count
  const int ksplit = argc > 4 ? atoi(argv[4]) : 0;
  const int launches = 64, g = 128;
  using PD = GemmPolicy<DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 1, 8, 1>;
  using PP = GemmPolicy<DK_INT4, LAYOUT_LOP3, AT_F16, MD_ZO, 0, 4, 4, 2, 0, true>;
  // 4th argument < 0: the K-sliced form of the decode member (plan suffix xdlk): 8 slices x min(fragments / 8, 32) groups
  const bool ksl = ksplit < 0;
  const int ksl_run = (((K / 128 + 7) / 8) + 3) & ~3;
  void (*fn)(const GemmArgs) = ksplit > 0 ? wq_gemm_kernel<PP> : ksl ? wq_gemm_decode_lds_kernel<PD, true> : wq_gemm_decode_lds_kernel<PD>;
  const int lds = ksplit > 0 ? 2 * 64 * 256 : ksl ? ksl_run * ((M + 3) / 4) * 1024 + 4096 : 8 * 4 * 16 * 256 + 8 * 64 * 16;
  const int nwaves = ksplit > 0 ? 4 : 8;
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
  const int tiles_m = ksplit > 0 ? (M + 63) / 64 : 1, tiles_n = ksplit > 0 ? (N + 127) / 128 : (N + 15) / 16;
  const int grid = ksl ? 8 * std::min((tiles_n + 7) / 8, 32) : tiles_m * tiles_n * (ksplit > 0 ? ksplit : 1);
  const size_t trace_words = (size_t)grid * nwaves * 16;
  void* WS = nullptr;
  if (ksplit > 1) CK(hipMalloc(&WS, (size_t)ksplit * M * N * 4));
  if (ksl) CK(hipMalloc(&WS, (size_t)tiles_n * 8 * 1024));
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
      a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.nsteps = K / 128; a.group_m = 1; a.ksplit = ksplit > 0 ? ksplit : 1; a.epi_tensor = 1.f;
      a.ws = WS;
      a.lut = T + (size_t)l * trace_words;
      hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * nwaves), lds, st, a);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rep %d: %.3f us/launch (eager, traced build)\n", rep, ms * 1000.f / launches);
  }
  std::vector<unsigned long long> t(trace_words * launches);
  CK(hipMemcpy(t.data(), T, t.size() * 8, hipMemcpyDeviceToHost));
  for (int l : {launches - 3, launches - 2, launches - 1}) {
    const unsigned long long* d = t.data() + (size_t)l * trace_words;
    const unsigned long long* dprev = t.data() + (size_t)(l - 1) * trace_words;
    const int nw = grid * nwaves;
    unsigned long long r0 = ~0ull, r1 = 0, p1 = 0;
    for (int w = 0; w < nw; ++w) { r0 = std::min(r0, d[w * 16 + 8]); r1 = std::max(r1, d[w * 16 + 9]); p1 = std::max(p1, dprev[w * 16 + 9]); }
    printf("launch %d: first wave start -> last wave end %.2f us; previous launch's last end -> this first start %.2f us\n", l, (r1 - r0) / 100.0,
           ((double)r0 - (double)p1) / 100.0);
    auto stat = [&](const char* name, auto f, double scale, const char* unit) {
      std::vector<double> v(nw);
      for (int w = 0; w < nw; ++w) v[w] = f(d + w * 16) * scale;
      std::sort(v.begin(), v.end());
      printf("  %-34s min %8.2f  p10 %8.2f  med %8.2f  p90 %8.2f  max %8.2f %s\n", name, v[0], v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1], unit);
    };
    stat("wave start since first start", [&](const unsigned long long* x) { return (double)(x[8] - r0); }, 0.01, "us");
    stat("wave end since first start", [&](const unsigned long long* x) { return (double)(x[9] - r0); }, 0.01, "us");
    stat("wave lifetime", [&](const unsigned long long* x) { return (double)(x[9] - x[8]); }, 0.01, "us");
    if (ksl) {
      stat("entry -> tile + 3 units asked for", [&](const unsigned long long* x) { return (double)(x[1] - x[0]); }, 1.0, "clk");
      stat("-> own share of the tile in LDS", [&](const unsigned long long* x) { return (double)(x[2] - x[1]); }, 1.0, "clk");
      stat("-> past the barrier", [&](const unsigned long long* x) { return (double)(x[3] - x[2]); }, 1.0, "clk");
      stat("-> first unit multiplied", [&](const unsigned long long* x) { return x[4] ? (double)(x[4] - x[3]) : 0.0; }, 1.0, "clk");
      stat("-> all units done", [&](const unsigned long long* x) { return x[4] ? (double)(x[5] - x[4]) : 0.0; }, 1.0, "clk");
      stat("entry -> all units done", [&](const unsigned long long* x) { return (double)(x[5] - x[0]); }, 1.0, "clk");
      stat("of which in the units' waits", [&](const unsigned long long* x) { return (double)x[6]; }, 1.0, "clk");
      stat("barrier -> done, not waiting", [&](const unsigned long long* x) { return (double)(x[5] - x[3]) - (double)x[6]; }, 1.0, "clk");
      continue;
    }
    if (ksplit > 0) {
      stat("entry -> first loads issued", [&](const unsigned long long* x) { return (double)(x[1] - x[0]); }, 1.0, "clk");
      stat("-> first tile in LDS (barrier)", [&](const unsigned long long* x) { return (double)(x[2] - x[1]); }, 1.0, "clk");
      stat("-> first k-step done (barrier)", [&](const unsigned long long* x) { return (double)(x[3] - x[2]); }, 1.0, "clk");
      stat("-> all k-steps done", [&](const unsigned long long* x) { return (double)(x[4] - x[3]); }, 1.0, "clk");
      stat("-> partial sums stored (issued)", [&](const unsigned long long* x) { return (double)(x[5] - x[4]); }, 1.0, "clk");
      stat("entry -> stores issued", [&](const unsigned long long* x) { return (double)(x[5] - x[0]); }, 1.0, "clk");
      continue;
    }
    stat("entry -> loads issued", [&](const unsigned long long* x) { return (double)(x[1] - x[0]); }, 1.0, "clk");
    stat("loads issued -> k-step 0 done", [&](const unsigned long long* x) { return (double)(x[2] - x[1]); }, 1.0, "clk");
    stat("k-step 0 done -> k-step 2 done", [&](const unsigned long long* x) { return (double)(x[3] - x[2]); }, 1.0, "clk");
    stat("k-step 2 done -> MFMAs done", [&](const unsigned long long* x) { return (double)(x[4] - x[3]); }, 1.0, "clk");
    stat("MFMAs done -> past barrier", [&](const unsigned long long* x) { return (double)(x[5] - x[4]); }, 1.0, "clk");
    stat("barrier -> store issued (wave 0)", [&](const unsigned long long* x) { return x[6] ? (double)(x[6] - x[5]) : 0.0; }, 1.0, "clk");
    stat("entry -> past barrier", [&](const unsigned long long* x) { return (double)(x[5] - x[0]); }, 1.0, "clk");
  }
  return 0;
}
