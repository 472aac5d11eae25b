// gemvs_lab_kernel.h - LAB ONLY: experimental GEMV members measured by tools/gemv_lab.hip against the library's launch.
#pragma once
#include "wqaa_gemvx_kernel.h"

template <class TimeIt>
static void run_gemvs_lab(int N, int K, int G, int count, void* dA, void* dS, std::vector<void*>& dW, int NSETS, std::vector<void*>& dC, void* dC2,
                          hipStream_t st, TimeIt&& time_it, double bytes) {
  (void)N; (void)K; (void)G; (void)count; (void)dA; (void)dS; (void)dW; (void)NSETS; (void)dC; (void)dC2; (void)st; (void)time_it; (void)bytes;
}
