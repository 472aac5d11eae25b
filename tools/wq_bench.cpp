// wq_bench.cpp - kernel-iteration harness: drives libwqaa_hip.so through its C ABI only.
//   wq_bench M N K [wfmt=1(int)] [bits=4] [group=128] [zeros_mode=0] [a_dtype=0(f16)] [rounds=4] [graph=1] [out_dtype]
// Rotates over enough distinct weight buffers to exceed the 256 MiB Infinity Cache, launches them
// back to back (eager or as one hipGraph), times the batch with hipEvents on the launch stream.
// Meant to be run under `rocprofv3 --kernel-trace --stats` as well.
// Build: hipcc -O2 -o tools/wq_bench tools/wq_bench.cpp -Iinclude -Lbitblas_amd -lwqaa_hip -Wl,-rpath,'$ORIGIN/../bitblas_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "wqaa.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static size_t dsize(int dt) { return dt == WQAA_F16 || dt == WQAA_BF16 ? 2 : dt == WQAA_F32 || dt == WQAA_I32 ? 4 : 1; }

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: wq_bench M N K [wfmt bits group zmode a_dtype rounds graph out_dtype]\n"); return 1; }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const int wfmt = argc > 4 ? atoi(argv[4]) : WQAA_W_INT;
  const int bits = argc > 5 ? atoi(argv[5]) : 4;
  const int group = argc > 6 ? atoi(argv[6]) : 128;
  const int zmode = argc > 7 ? atoi(argv[7]) : 0;
  const int adt = argc > 8 ? atoi(argv[8]) : WQAA_F16;
  const int rounds = argc > 9 ? atoi(argv[9]) : 4;
  const int use_graph = argc > 10 ? atoi(argv[10]) : 1;
  const int odt = argc > 11 ? atoi(argv[11]) : (adt == WQAA_I8 ? WQAA_I32 : WQAA_F16);
  init();
  wqaa_matmul_desc d; memset(&d, 0, sizeof d);
  d.struct_size = sizeof d; d.N = N; d.K = K; d.a_dtype = adt; d.w_format = wfmt; d.w_bits = bits;
  d.out_dtype = odt; d.group_size = group; d.with_scaling = (adt == WQAA_F16 && wfmt != WQAA_W_NATIVE && getenv("WQ_NOSCALE") == nullptr) ? 1 : 0;
  d.zeros_mode = zmode; d.with_bias = 0; d.w_layout = (wfmt <= WQAA_W_INT && bits < 8 && getenv("WQ_PLAIN") == nullptr) ? WQAA_LAYOUT_LOP3 : WQAA_LAYOUT_PLAIN;
  d.strict_reference = getenv("WQ_STRICT") ? atoi(getenv("WQ_STRICT")) : 1;
  wqaa_plan plan;
  if (wqaa_select(&d, M, &plan) != WQAA_OK) { printf("select failed: %s\n", wqaa_last_error_string()); return 2; }
  const int g = group <= 0 ? K : group;
  const size_t wbytes = (size_t)N * K * bits / 8;
  const size_t sbytes = (size_t)N * (K / g) * 2;
  const size_t zbytes = zmode == WQAA_Z_QUANTIZED ? (size_t)(K / g) * N * bits / 8 : sbytes;
  const size_t abytes = (size_t)M * K * dsize(adt), cbytes = (size_t)M * N * dsize(odt);
  const double alg = (double)abytes + wbytes + (d.with_scaling ? sbytes : 0) + (zmode ? zbytes : 0) + cbytes;
  int nbuf = (int)std::max<size_t>(2, std::min<size_t>(256, (640ull << 20) / wbytes));
  printf("%s\n  M=%d N=%d K=%d bits=%d g=%d zmode=%d grid=%d threads=%d lds=%d  alg_bytes=%.0f  nbuf=%d (%.0f MB)\n", plan.name, M, N, K, bits, g, zmode,
         plan.grid, plan.threads, plan.lds_bytes, alg, nbuf, nbuf * wbytes / 1e6);
  std::vector<uint8_t> h(std::max(wbytes, std::max(abytes, sbytes)));
  srand(1);
  for (auto& b : h) b = (uint8_t)rand();
  std::vector<void*> W(nbuf), S(nbuf), Z(nbuf);
  for (int i = 0; i < nbuf; ++i) {
    CK(hipMalloc(&W[i], wbytes)); CK(hipMemcpy(W[i], h.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&S[i], sbytes)); CK(hipMalloc(&Z[i], zbytes));
  }
  // scales: small fp16 values (0x2xxx ~ 0.01), zeros: 8.0 (0x4800) or packed 0x88
  std::vector<uint16_t> hs(sbytes / 2); for (auto& v : hs) v = 0x2000 | (rand() & 0x3ff);
  std::vector<uint16_t> hz(zbytes / 2 + 1, zmode == WQAA_Z_QUANTIZED ? 0x8888 : 0x4800);
  for (int i = 0; i < nbuf; ++i) { CK(hipMemcpy(S[i], hs.data(), sbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(Z[i], hz.data(), zbytes, hipMemcpyHostToDevice)); }
  void *A, *C; CK(hipMalloc(&A, abytes)); CK(hipMalloc(&C, cbytes));
  if (adt == WQAA_F16) { std::vector<uint16_t> ha(abytes / 2); for (auto& v : ha) v = (rand() & 1 ? 0x8000 : 0) | 0x3000 | (rand() & 0x7ff); CK(hipMemcpy(A, ha.data(), abytes, hipMemcpyHostToDevice)); }
  else CK(hipMemcpy(A, h.data(), abytes, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  auto launch_all = [&]() {
    for (int i = 0; i < nbuf; ++i) {
      int st = wqaa_matmul(&d, A, W[i], nullptr, d.with_scaling ? S[i] : nullptr, zmode ? Z[i] : nullptr, nullptr, C, M, s);
      if (st != WQAA_OK) { printf("launch failed: %s\n", wqaa_last_error_string()); exit(3); }
    }
  };
  launch_all(); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraphExec_t ge = nullptr;
  if (use_graph) {
    hipGraph_t gr; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); launch_all(); CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  }
  std::vector<float> per;
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0, s));
    if (use_graph) CK(hipGraphLaunch(ge, s)); else launch_all();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); per.push_back(ms * 1e3f / nbuf);
  }
  std::sort(per.begin(), per.end());
  const double us = per[per.size() / 2];
  printf("  %s: %.3f us/launch (median of %d rounds; min %.3f)  -> %.1f GB/s algorithmic, %.2f TFLOP/s\n", use_graph ? "graph" : "eager", us, rounds, per[0],
         alg / us * 1e-3, 2.0 * M * N * K / us * 1e-6);
  return 0;
}
