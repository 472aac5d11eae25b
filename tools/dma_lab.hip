// dma_lab.hip - what one loader wave per CU can stream into LDS with global_load_lds_dwordx4 on MI355X, by address pattern,
// cache policy, queue depth and loader count (the weight stream of csrc/wqaa_chain_kernel.h, nothing else in the kernel).
//   hipcc --offload-arch=gfx950 -O3 -o tools/dma_lab tools/dma_lab.hip && tools/dma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT, bool VADDR>
__device__ __forceinline__ void dma(unsigned lds_dst, unsigned voff, unsigned long long sbase) {
  unsigned keep;
  if (VADDR) {
    const unsigned long long a = sbase + voff;
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(a), "s"(lds_dst) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(a), "s"(lds_dst) : "memory");
  } else {
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
  }
}

// pattern 0: CU b streams its own contiguous slice [b * units .. (b + 1) * units) KiB
//         1: KiB u of CU b = (u * G + b): the chip walks ONE contiguous window
//         2: blocks of 16 KiB round-robin over the CUs
//         3: as 0 but every CU starts at another offset inside its slice (rotated by b * 7 units)
template <bool NT, bool VADDR, int LAG>
__global__ void __launch_bounds__(256) k_stream(const unsigned char* base, int units, int pattern, int loaders, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x, G = gridDim.x;
  if (wave >= loaders) return;
  const unsigned ring_units = 96 / loaders;
  const unsigned ring0 = wave * ring_units * 1024;
  unsigned rpos = 0;
  int inflight = 0;
  for (int u = wave; u < units; u += loaders) {
    long kib;
    if (pattern == 0) kib = (long)b * units + u;
    else if (pattern == 1) kib = (long)u * G + b;
    else if (pattern == 2) kib = ((long)(u >> 4) * G + b) * 16 + (u & 15);
    else { int uu = u + b * 7; uu %= units; kib = (long)b * units + uu; }
    const unsigned long long src = (unsigned long long)base + (unsigned long long)kib * 1024ull;
    dma<NT, VADDR>(ring0 + rpos * 1024, lane * 16, src);
    if (++rpos == ring_units) rpos = 0;
    if (++inflight == 8) {
      inflight = 0;
      if (LAG == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      if (LAG == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      if (LAG == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (LAG == 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && b == 0) sink[wave] = ((unsigned*)smem)[ring0 / 4];
}

typedef void (*kfn)(const unsigned char*, int, int, int, unsigned*);

int main() {
  const int G = 256, units = 300;                 // 300 KiB per CU = one decoder-layer tail
  const size_t bytes = (size_t)G * units * 1024;
  const int nbuf = 8;                             // rotate over 8 x 75 MiB: past the 256 MiB memory-side cache
  std::vector<unsigned char*> bufs(nbuf);
  for (auto& p : bufs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
  unsigned* sink;
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { const char* name; kfn fn; } vs[] = {
    {"saddr nt lag48", k_stream<true, false, 48>}, {"saddr default lag48", k_stream<false, false, 48>},
    {"vaddr nt lag48", k_stream<true, true, 48>}, {"saddr nt lag32", k_stream<true, false, 32>},
    {"saddr nt lag16", k_stream<true, false, 16>}, {"saddr nt lag56", k_stream<true, false, 56>},
  };
  for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-22s %-8s %-8s %10s %10s\n", "variant", "pattern", "loaders", "us", "TB/s");
  for (auto& v : vs)
    for (int pattern = 0; pattern < 4; ++pattern)
      for (int loaders : {1, 2, 4}) {
        if (loaders > 1 && &v != &vs[0]) continue;
        if (pattern > 1 && &v != &vs[0]) continue;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(v.fn, dim3(G), dim3(256), 98304, 0, bufs[w], units, pattern, loaders, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 16;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.fn, dim3(G), dim3(256), 98304, 0, bufs[r % nbuf], units, pattern, loaders, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-22s %-8d %-8d %10.2f %10.3f\n", v.name, pattern, loaders, us, bytes / us / 1e6);
      }
  return 0;
}
