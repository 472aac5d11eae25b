// floor_bench.hip - measurement floor for the M=1 GEMV on MI355X: how long does ANY kernel take to
// pull 8.39 MB (a 4096x4096 int4 matrix) from HBM, and what does an empty launch cost, when timed
// by the kernel's own begin/end timestamps (hipExtLaunchKernel events; same clock rocprofv3 uses).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/floor_bench tools/floor_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(int* out) { if (out && threadIdx.x == 9999) out[0] = 1; }

// each block reads a contiguous slab: U loads of 16 B per lane issued before any use
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, long n_vec, uint32_t* out) {
  const long per_block = (long)blockDim.x * U;
  uint32_t acc = 0;
  for (long base = (long)blockIdx.x * per_block; base < n_vec; base += (long)gridDim.x * per_block) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * blockDim.x + threadIdx.x;
      if (NT) v[u] = __builtin_nontemporal_load(p + (i < n_vec ? i : 0));
      else v[u] = p[i < n_vec ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static void run(const char* name, F launch, int nbuf, int reps, double bytes) {
  std::vector<hipEvent_t> e0(nbuf), e1(nbuf);
  for (int i = 0; i < nbuf; ++i) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
  std::vector<float> d;
  for (int r = 0; r <= reps; ++r) {
    for (int i = 0; i < nbuf; ++i) launch(i, e0[i], e1[i]);
    CK(hipDeviceSynchronize());
    if (r == 0) continue;
    for (int i = 0; i < nbuf; ++i) { float ms; CK(hipEventElapsedTime(&ms, e0[i], e1[i])); d.push_back(ms * 1e3f); }
  }
  std::sort(d.begin(), d.end());
  double mean = 0; for (float x : d) mean += x; mean /= d.size();
  printf("%-40s mean %7.3f us  median %7.3f  min %7.3f  p90 %7.3f", name, mean, d[d.size() / 2], d[0], d[d.size() * 9 / 10]);
  if (bytes > 0) printf("   -> %7.1f GB/s (mean) %7.1f (min)", bytes / mean * 1e-3, bytes / d[0] * 1e-3);
  printf("\n");
  for (int i = 0; i < nbuf; ++i) { hipEventDestroy(e0[i]); hipEventDestroy(e1[i]); }
}

// batch timing: plain hipEventRecord around `n` back-to-back launches (includes inter-kernel gaps)
template <class F>
static void run_batch(const char* name, F launch, int nbuf, int rounds, double bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipStream_t s0 = nullptr;
  for (int i = 0; i < nbuf; ++i) launch(i, s0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, s0));
  for (int r = 0; r < rounds; ++r) for (int i = 0; i < nbuf; ++i) launch(i, s0);
  CK(hipEventRecord(b, s0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / (rounds * nbuf);
  printf("%-40s batch %7.3f us/launch", name, us);
  if (bytes > 0) printf("   -> %7.1f GB/s", bytes / us * 1e-3);
  printf("\n");
}

int main(int argc, char** argv) {
  const long bytes = argc > 1 ? atol(argv[1]) : 4096L * 4096 / 2;
  const int nbuf = (int)std::max(2L, (600L << 20) / bytes);
  printf("buffer %ld bytes, %d rotating buffers (%.0f MB)\n", bytes, nbuf, (double)nbuf * bytes / 1e6);
  std::vector<u32x4*> buf(nbuf);
  for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], i + 1, bytes)); }
  uint32_t* out; CK(hipMalloc(&out, 64));
  hipStream_t s; CK(hipStreamCreate(&s));
  const long n_vec = bytes / 16;
  for (int grid : {256, 512, 1024}) {
    char nm[64]; snprintf(nm, sizeof nm, "empty grid=%d", grid);
    run(nm, [&](int, hipEvent_t a, hipEvent_t b) {
      int* o = nullptr; void* args[] = {&o};
      CK(hipExtLaunchKernel((const void*)k_empty, dim3(grid), dim3(256), args, 0, s, a, b, 0)); }, nbuf, 3, 0);
  }
#define RD(U, NT, GRID) { char nm[64]; snprintf(nm, sizeof nm, "read U=%d nt=%d grid=%d", U, NT, GRID); \
  run(nm, [&](int i, hipEvent_t a, hipEvent_t b) { const u32x4* p = buf[i]; long nv = n_vec; uint32_t* o = out; void* args[] = {&p, &nv, &o}; \
    CK(hipExtLaunchKernel((const void*)k_read<U, NT>, dim3(GRID), dim3(256), args, 0, s, a, b, 0)); }, nbuf, 3, (double)bytes); }
  const int g1 = (int)((n_vec + 256 * 4 - 1) / (256 * 4)), g2 = (int)((n_vec + 256 * 8 - 1) / (256 * 8)), g0 = (int)((n_vec + 256 * 2 - 1) / (256 * 2)), g3 = (int)((n_vec + 256 * 16 - 1) / (256 * 16));
  RD(2, true, g0) RD(4, true, g1) RD(8, true, g2) RD(16, true, g3)
  RD(2, false, g0) RD(4, false, g1) RD(8, false, g2) RD(16, false, g3)
  RD(4, true, 256) RD(4, true, 512) RD(4, true, 1024) RD(8, true, 256) RD(8, true, 512) RD(16, true, 256)
  run_batch("empty grid=512", [&](int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st, (int*)nullptr); }, nbuf, 4, 0);
  run_batch("read U=4 nt=1 grid=512", [&](int i, hipStream_t st) { hipLaunchKernelGGL((k_read<4, true>), dim3(512), dim3(256), 0, st, (const u32x4*)buf[i], n_vec, out); }, nbuf, 4, (double)bytes);
  run_batch("read U=8 nt=1 grid=256", [&](int i, hipStream_t st) { hipLaunchKernelGGL((k_read<8, true>), dim3(256), dim3(256), 0, st, (const u32x4*)buf[i], n_vec, out); }, nbuf, 4, (double)bytes);
  run_batch("read U=4 nt=1 full grid", [&](int i, hipStream_t st) { hipLaunchKernelGGL((k_read<4, true>), dim3(g1), dim3(256), 0, st, (const u32x4*)buf[i], n_vec, out); }, nbuf, 4, (double)bytes);
  run_batch("read U=2 nt=1 full grid", [&](int i, hipStream_t st) { hipLaunchKernelGGL((k_read<2, true>), dim3(g0), dim3(256), 0, st, (const u32x4*)buf[i], n_vec, out); }, nbuf, 4, (double)bytes);
  return 0;
}
