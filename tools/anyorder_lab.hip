// anyorder_lab.hip - does a kernel launched with hipExtAnyOrderLaunch start before its predecessor on the same stream has ended?
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this asks the box.)  Two launches of a kernel whose workgroups
// stamp s_memrealtime at start and end around a spin of `work` ticks; A with a plain launch, B with the flag (or without: control).
//   hipcc --offload-arch=gfx950 -O3 tools/anyorder_lab.hip -o tools/anyorder_lab && tools/anyorder_lab
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

__global__ void stamp_kernel(unsigned long long* out, int work) {
  // an uneven kernel: the first 64 workgroups run four times as long - a tail during which most of the chip is free
  if (blockIdx.x < 64) work *= 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < (unsigned long long)work) {
    __builtin_amdgcn_s_sleep(8);
    t = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = t;
  }
}

static void span(const std::vector<unsigned long long>& v, unsigned long long* lo, unsigned long long* hi_start, unsigned long long* hi) {
  *lo = ~0ull;
  *hi = 0;
  *hi_start = 0;
  for (size_t i = 0; i < v.size(); i += 2) {
    *lo = std::min(*lo, v[i]);
    *hi_start = std::max(*hi_start, v[i]);
    *hi = std::max(*hi, v[i + 1]);
  }
}

int main() {
  const int grid = 2048, block = 256, work = 500;      // 500 ticks of 10 ns = 5 us per workgroup; 2048 workgroups: two rounds of 4 per CU
  unsigned long long *dA, *dB;
  CK(hipMalloc(&dA, grid * 16));
  CK(hipMalloc(&dB, grid * 16));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::vector<unsigned long long> hA(grid * 2), hB(grid * 2);
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: B plain; 1: B any-order (eager); 2: both captured into a graph, B launched any-order inside the capture
    for (int rep = 0; rep < 3; ++rep) {
      void* pa[] = {&dA, (void*)&work};
      void* pb[] = {&dB, (void*)&work};
      hipGraph_t g = nullptr;
      hipGraphExec_t ge = nullptr;
      if (mode == 2) CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      CK(hipExtLaunchKernel((const void*)stamp_kernel, dim3(grid), dim3(block), pa, 0, s, nullptr, nullptr, 0));
      CK(hipExtLaunchKernel((const void*)stamp_kernel, dim3(grid), dim3(block), pb, 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0));
      if (mode == 2) {
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
      }
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(hA.data(), dA, grid * 16, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hB.data(), dB, grid * 16, hipMemcpyDeviceToHost));
      unsigned long long a0, a1s, a1, b0, b1s, b1;
      span(hA, &a0, &a1s, &a1);
      span(hB, &b0, &b1s, &b1);
      printf("mode %d (%s) rep %d: A [0 .. last start %.2f .. end %.2f] us   B first start %.2f  last start %.2f  end %.2f   B starts %+.2f us after A's end\n", mode,
             mode == 0 ? "plain" : mode == 1 ? "any-order eager" : "any-order captured", rep, (a1s - a0) * 0.01, (a1 - a0) * 0.01, (b0 - a0) * 0.01,
             (b1s - a0) * 0.01, (b1 - a0) * 0.01, ((double)b0 - (double)a1) * 0.01);
      if (ge) CK(hipGraphExecDestroy(ge));
      if (g) CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
