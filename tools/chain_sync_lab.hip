// chain_sync_lab.hip - the fixed costs inside csrc/wqaa_chain_kernel.h's staging: a meeting of the consumer waves through an LDS
// counter, one in-place chunk staging, the norm's partial sums - shader cycles, isolated (one workgroup per CU, 16 waves).
#include "wqaa_chain_kernel.h"
#include <cstdio>
#include <vector>
using namespace wqaa;
using P = GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_sync(int rounds, int mode, unsigned long long* out, uint32_t* ctl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0u;
  __syncthreads();
  ChainArgs fake;
  ChainWave cw;
  cw.smem = smem; cw.args = nullptr; cw.lane = lane; cw.wave = wave; cw.b = blockIdx.x; cw.G = gridDim.x; cw.timeout = 1u << 30;
  (void)fake;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    if (mode == 0) {                       // a meeting: arrive + wait for everyone
      CHAIN_LDS_RELEASE();
      if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(smem) + 64 + (r & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unsigned n = 0;
      while ((int)chain_lds_ld(smem, 64 + (r & 1)) < (r / 2 + 1) * nw) { __builtin_amdgcn_s_sleep(1); ++n; }
      CHAIN_LDS_ACQUIRE();
    } else if (mode == 1) {                // one chunk staged in place (this wave's own region)
      chain_stage_chunk<P, false>(smem, 4096, 4096 + 16 * 4096, 1 << 20, wave, lane, 0.f, nullptr);
    } else if (mode == 2) {                // s_memrealtime + a store: the cost of a time stamp
      if (lane == 0) out[1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    } else if (mode == 3) {                // an agent-scope relaxed load of one word (the generation)
      const uint32_t g = __hip_atomic_load((chain_gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g == 12345u) chain_lds_st(smem, 70, g);
    } else if (mode == 4) {                // a poll of an LDS word that is already there
      if (!cw.wait_ge(64, 0u, 1, 0)) return;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
  unsigned long long* dout;
  uint32_t* ctl;
  CK(hipMalloc(&dout, 8 * 8192));
  CK(hipMalloc(&ctl, 256));
  CK(hipMemset(ctl, 0, 256));
  CK(hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const char* names[] = {"meeting of the waves (LDS counter)", "chunk staged in place", "time stamp (s_memrealtime + store)", "agent-scope load of one word", "poll of a ready LDS word"};
  for (int mode = 0; mode < 5; ++mode)
    for (int rounds : {64, 1})
    for (int waves : {4, 12, 16}) {
      const int G = 256;
      if (mode == 2) continue;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_sync, dim3(G), dim3(64 * waves), 100 * 1024, 0, rounds, mode, dout, ctl);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(G * 16);
      CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0;
      int cnt = 0;
      for (int b = 0; b < G; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 16 + w]; ++cnt; }
      printf("%-40s %2d waves/CU, %2d round(s) per launch: %8.1f cycles per round per wave\n", names[mode], waves, rounds, sum / cnt / rounds);
    }
  return 0;
}
