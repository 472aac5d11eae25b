// issue_lab.hip - lab (round 5): what a SIMD of gfx950 does with the decode members' instruction mix.  The K-sliced decode form's time
// line (profiles/r05_kslice_trace.txt) shows waves that never wait for memory in their unit loop and still need ~470 clocks per k-step
// and SIMD (2 waves): 84 vector instructions (336 clocks of issue) + 4 dependent v_mfma_f32_16x16x32_f16 (128 clocks of matrix pipe).
// Questions: (1) does a chain of DEPENDENT MFMAs (same accumulator) run at the rate of independent ones?  (2) do the VALU stream of one wave
// and the MFMA stream of another wave on the same SIMD overlap, or add?  (3) the same inside ONE wave (MFMAs followed by independent VALU).
// One workgroup of 8 waves per CU slot (waves w and w + 4 share a SIMD); clocks per loop trip from s_memtime around the loop.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/issue_lab tools/issue_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bits per wave role: 1 = MFMA stream (dependent chain of 4), 2 = MFMA stream (two accumulators), 4 = VALU stream (84 pk ops per trip),
// 8 = both in one wave (4 dependent MFMAs, then 84 VALU independent of them)
__global__ void __launch_bounds__(512) k_issue(int role_lo, int role_hi, int trips, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = wave < 4 ? role_lo : role_hi;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  half2v v[12];
  for (int i = 0; i < 12; ++i) v[i] = half2v{(_Float16)(lane + i), (_Float16)(i * 0.5f)};
  const half2v c1 = {(_Float16)1.0009765625f, (_Float16)0.99951171875f}, c2 = {(_Float16)0.001f, (_Float16)-0.001f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
    if (role & 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
    }
    if (role & 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
    }
    if (role & 8) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
    }
    if (role & (4 | 8)) {
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          v[i] = v[i] * c1 + c2;                 // (-ffp-contract=off is not set: one v_pk_fma_f16 per element pair)
          asm volatile("" : "+v"(v[i]));
        }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  float s = acc0[0] + acc1[1];
  for (int i = 0; i < 12; ++i) s += (float)v[i][0];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

static void run(const char* name, int lo, int hi, unsigned long long* d, float* sink) {
  const int trips = 2000;
  hipLaunchKernelGGL(k_issue, dim3(256), dim3(512), 0, 0, lo, hi, trips, d, sink);
  CK(hipDeviceSynchronize());
  unsigned long long h[8];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-58s waves 0-3: %7.1f clk/trip   waves 4-7: %7.1f clk/trip\n", name, (double)h[0] / trips, (double)h[4] / trips);
}

int main() {
  unsigned long long* d; float* sink;
  CK(hipMalloc(&d, 256 * 8 * 8)); CK(hipMalloc(&sink, 4096));
  run("warm-up", 1, 4, d, sink);
  run("4 dependent MFMAs | idle", 1, 0, d, sink);
  run("4 MFMAs on two accumulators | idle", 2, 0, d, sink);
  run("84 VALU | idle", 4, 0, d, sink);
  run("4 dependent MFMAs + 84 VALU in ONE wave | idle", 8, 0, d, sink);
  run("4 dependent MFMAs | 84 VALU (other wave, same SIMD)", 1, 4, d, sink);
  run("MFMAs + VALU in one wave | the same in the other", 8, 8, d, sink);
  run("4 dep. MFMAs | 4 dep. MFMAs", 1, 1, d, sink);
  run("84 VALU | 84 VALU", 4, 4, d, sink);
  return 0;
}
