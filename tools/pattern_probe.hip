// pattern_probe.hip - does the lane -> address mapping of the small-M MFMA members cost HBM efficiency?
// Two read-only kernels stream the same N x RB byte matrix (rows of RB bytes) with the same workgroup
// shape and the same number of 16-byte loads in flight per lane:
//   row  : a wave instruction reads 1 KiB contiguous of ONE row (the GEMV family's mapping)
//   frag : a wave instruction reads 64 B of each of 16 rows (lane = (row & 15, k-block), the MFMA operand
//          mapping of the GEMM family: weights go straight from memory into matrix-core operands)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pattern_probe tools/pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// workgroup = 4 waves; it covers 64 rows x (U * 64) bytes ("frag") or 16 rows x (U * 256) bytes ("row"):
// U loads per lane, all issued before use, 16 KiB per workgroup when U = 4
template <int U, bool FRAG>
__global__ void __launch_bounds__(256) k_stream(const uint8_t* __restrict__ p, int N, long RB, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  if (FRAG) {
    const long chunks_per_row = RB / (U * 64);
    const long ntile = (long)(N / 64) * chunks_per_row;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
      const long rt = t / chunks_per_row, ck = t % chunks_per_row;
      const long row = rt * 64 + wave * 16 + (lane & 15);
      const uint8_t* q = p + row * RB + ck * (U * 64) + (lane >> 4) * 16;
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q + u * 64));
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  } else {
    const long chunks_per_row = RB / (U * 256);
    const long ntile = (long)(N / 16) * chunks_per_row;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
      const long rt = t / chunks_per_row, ck = t % chunks_per_row;
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long row = rt * 16 + wave * 4 + u;      // 4 rows per wave, 1 KiB contiguous each... per chunk of U*256 B
        const uint8_t* q = p + row * RB + ck * (U * 256) + (long)lane * 16;
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static double time_graph(F launch, int reps) {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < reps; ++i) launch(i, s);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> d;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); d.push_back(ms * 1e3f / reps);
  }
  std::sort(d.begin(), d.end());
  return d[d.size() / 2];
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 8192;
  const long RB = argc > 2 ? atol(argv[2]) : 14336;     // bytes per row (K * bits / 8)
  const int nbuf = 4;
  std::vector<uint8_t*> bufs(nbuf);
  for (auto& b : bufs) { CK(hipMalloc(&b, (size_t)N * RB)); CK(hipMemset(b, 1, (size_t)N * RB)); }
  uint32_t* out; CK(hipMalloc(&out, 4));
  const double bytes = (double)N * RB;
  printf("matrix %d x %ld B = %.1f MB\n", N, RB, bytes / 1e6);
  for (int grid : {2048, 8192, 1 << 20}) {
    const double tr = time_graph([&](int i, hipStream_t s) { hipLaunchKernelGGL((k_stream<4, false>), dim3(grid), dim3(256), 0, s, bufs[i % nbuf], N, RB, out); }, 8);
    const double tf = time_graph([&](int i, hipStream_t s) { hipLaunchKernelGGL((k_stream<4, true>), dim3(grid), dim3(256), 0, s, bufs[i % nbuf], N, RB, out); }, 8);
    const double tf16 = time_graph([&](int i, hipStream_t s) { hipLaunchKernelGGL((k_stream<16, true>), dim3(grid), dim3(256), 0, s, bufs[i % nbuf], N, RB, out); }, 8);
    printf("grid %7d: row-contiguous %7.2f us (%5.2f TB/s)   fragment (16 rows x 64 B) %7.2f us (%5.2f TB/s)   fragment, 16 loads/lane %7.2f us (%5.2f TB/s)\n",
           grid, tr, bytes / tr * 1e-6, tf, bytes / tf * 1e-6, tf16, bytes / tf16 * 1e-6);
  }
  return 0;
}
