// kernarg_lab.hip - what does a wave pay for its kernel arguments?  Time from a wave's first instruction to its arguments being in
// SGPRs (s_memrealtime, 100 MHz), (a) a by-value struct fetched with s_load (what every member of this library does), (b) the same
// values as leading scalar arguments PRELOADED into user SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count=N,
// gfx940+), inside a captured hipGraph of many launches - the bench's launch path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 tools/kernarg_lab.hip -o tools/kernarg_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const uint32_t* a;
  uint32_t* out;
  unsigned long long* stamps;
  int n, k, x0, x1, x2, x3, x4, x5;
  int pad[40];      // (a group launch's argument block is ~1 KB: the member of interest sits somewhere inside)
};

__global__ void __launch_bounds__(512) k_struct(const Args s) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"s"(s.a), "s"(s.out), "s"(s.stamps), "s"(s.n), "s"(s.k), "s"(s.x0), "s"(s.x1), "s"(s.x2), "s"(s.x3));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t v = s.a[(threadIdx.x + s.k + s.x0 + s.x1 + s.x2 + s.x3) & (s.n - 1)];
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { s.stamps[w * 3] = t0; s.stamps[w * 3 + 1] = t1; s.stamps[w * 3 + 2] = t2; }
  if (v == 0xdeadbeefu) s.out[0] = v;
}

__global__ void __launch_bounds__(512) k_scalar(const uint32_t* a, uint32_t* out, unsigned long long* stamps, int n, int k, int x0, int x1, int x2, int x3) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"s"(a), "s"(out), "s"(stamps), "s"(n), "s"(k), "s"(x0), "s"(x1), "s"(x2), "s"(x3));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t v = a[(threadIdx.x + k + x0 + x1 + x2 + x3) & (n - 1)];
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { stamps[w * 3] = t0; stamps[w * 3 + 1] = t1; stamps[w * 3 + 2] = t2; }
  if (v == 0xdeadbeefu) out[0] = v;
}

int main() {
  const int grid = 512, threads = 512, nl = 16, waves = grid * threads / 64;
  uint32_t *a, *out;
  unsigned long long* st;
  CK(hipMalloc(&a, 1 << 20));
  CK(hipMemset(a, 1, 1 << 20));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&st, (size_t)nl * waves * 3 * 8));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int mode = 0; mode < 2; ++mode) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < nl; ++l) {
      if (mode == 0) {
        Args x{};
        x.a = a; x.out = out; x.stamps = st + (size_t)l * waves * 3; x.n = 1 << 18; x.k = l; x.x0 = 1; x.x1 = 2; x.x2 = 3; x.x3 = 4;
        hipLaunchKernelGGL(k_struct, dim3(grid), dim3(threads), 0, s, x);
      } else {
        hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(threads), 0, s, a, out, st + (size_t)l * waves * 3, 1 << 18, l, 1, 2, 3, 4);
      }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)nl * waves * 3);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> arg, ld, first;
    for (int l = 0; l < nl; ++l) {
      unsigned long long tmin = ~0ull;
      for (int w = 0; w < waves; ++w) tmin = std::min(tmin, h[((size_t)l * waves + w) * 3]);
      double fmin = 1e30;
      for (int w = 0; w < waves; ++w) {
        const unsigned long long* p = &h[((size_t)l * waves + w) * 3];
        arg.push_back((p[1] - p[0]) * 0.01);
        ld.push_back((p[2] - p[1]) * 0.01);
        fmin = std::min(fmin, (double)(p[1] - tmin) * 0.01);
      }
      first.push_back(fmin);
    }
    std::sort(arg.begin(), arg.end()); std::sort(ld.begin(), ld.end()); std::sort(first.begin(), first.end());
    printf("%-44s launch %.2f us | wave entry -> arguments in SGPRs: median %.2f us  p10 %.2f  p90 %.2f  max %.2f | then one load: median %.2f | earliest 'arguments known' after the launch's first wave: median %.2f\n",
           mode == 0 ? "by-value struct (s_load)" : "leading scalars (preload where supported)", ms * 1e3 / 20 / nl, arg[arg.size() / 2],
           arg[arg.size() / 10], arg[arg.size() * 9 / 10], arg.back(), ld[ld.size() / 2], first[first.size() / 2]);
  }
  return 0;
}
