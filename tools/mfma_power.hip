// mfma_power.hip - what the matrix pipe sustains on this chip under its power limit, per instruction shape and operand data:
// every SIMD holds two waves that do nothing but MFMAs on register operands (8 or 16 independent accumulators), optionally
// with one ds_read_b128 per 32 matrix-pipe cycles.  Prints TFLOP/s (TOP/s) and the clock the run sustained.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// MODE 0: f16 32x32x16, 1: f16 16x16x32, 2: i8 32x32x32, 3: i8 16x16x64; LDS: one ds_read_b128 per 32 pipe cycles feeds the B operand
template <int MODE, bool LDS>
__global__ void __launch_bounds__(512) k(const u32x4* in, float* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  u32x4 a = in[threadIdx.x], b[8];
  for (int i = 0; i < 8; ++i) b[i] = in[512 + ((threadIdx.x * 8 + i) & 4095)];
  for (int i = threadIdx.x; i < 8192; i += 512) reinterpret_cast<u32x4*>(smem)[i] = in[i & 4095];
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE == 0 || MODE == 2) {
    typename std::conditional<MODE == 0, f32x16, i32x16>::type acc[8];
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 16; ++i) acc[f][i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        u32x4 bb = b[f];
        if constexpr (LDS) bb = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + f * 4096 + (it & 7) * 1024) & 131071));
        if constexpr (MODE == 0) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bb), acc[f], 0, 0, 0);
        else acc[f] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, bb), acc[f], 0, 0, 0);
      }
    }
    float s = 0;
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 16; ++i) s += (float)acc[f][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    typename std::conditional<MODE == 1, f32x4, i32x4>::type acc[16];
    for (int f = 0; f < 16; ++f) for (int i = 0; i < 4; ++i) acc[f][i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        u32x4 bb = b[f & 7];
        if constexpr (LDS) { if ((f & 1) == 0) bb = *reinterpret_cast<const u32x4*>(smem + ((lane * 16 + f * 2048 + (it & 7) * 1024) & 131071)); }
        if constexpr (MODE == 1) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, bb), acc[f], 0, 0, 0);
        else acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, bb), acc[f], 0, 0, 0);
      }
    }
    float s = 0;
    for (int f = 0; f < 16; ++f) for (int i = 0; i < 4; ++i) s += (float)acc[f][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

// fp8 (e4m3): FM 0 = 16x16x32 plain, 1 = 16x16x128 block-scaled with unit scales, 2 = 32x32x64 block-scaled
template <int FM>
__global__ void __launch_bounds__(512) k8(const u32x4* in, float* out, int iters, unsigned long long* clk) {
  u32x4 a0 = in[threadIdx.x], a1 = in[1024 + threadIdx.x], b0[8], b1[8];
  for (int i = 0; i < 8; ++i) { b0[i] = in[512 + ((threadIdx.x * 8 + i) & 4095)]; b1[i] = in[2048 + ((threadIdx.x * 8 + i) & 4095)]; }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  if constexpr (FM == 2) {
    f32x16 acc[8];
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 16; ++i) acc[f][i] = 0;
    const i32x8 av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const i32x8 bv = {(int)b0[f][0], (int)b0[f][1], (int)b0[f][2], (int)b0[f][3], (int)b1[f][0], (int)b1[f][1], (int)b1[f][2], (int)b1[f][3]};
        acc[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[f], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    }
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 16; ++i) s += acc[f][i];
  } else {
    f32x4 acc[16];
    for (int f = 0; f < 16; ++f) for (int i = 0; i < 4; ++i) acc[f][i] = 0;
    const i32x8 av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        if constexpr (FM == 1) {
          const i32x8 bv = {(int)b0[f & 7][0], (int)b0[f & 7][1], (int)b0[f & 7][2], (int)b0[f & 7][3], (int)b1[f & 7][0], (int)b1[f & 7][1], (int)b1[f & 7][2], (int)b1[f & 7][3]};
          acc[f] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[f], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        } else {
          const long al = ((long)a0[1] << 32) | a0[0], bl = ((long)b0[f & 7][1] << 32) | b0[f & 7][0];
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc[f], 0, 0, 0);
        }
      }
    }
    for (int f = 0; f < 16; ++f) for (int i = 0; i < 4; ++i) s += acc[f][i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int FM>
static void run8(const char* name, const u32x4* din, float* dout, unsigned long long* dclk, int iters) {
  auto fn = k8<FM>;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 0, 0, din, dout, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[512];
  CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
  const double per = FM == 0 ? 16.0 * 2 * 16 * 16 * 32 : FM == 1 ? 16.0 * 2 * 16 * 16 * 128 : 8.0 * 2 * 32 * 32 * 64;
  const double flop = 20.0 * 256 * 8 * (double)iters * per;
  printf("%-28s %8.1f T/s   clock %.3f GHz (block 0)\n", name, flop / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] * 10.0));
}

template <int MODE, bool LDS>
static void run(const char* name, const u32x4* din, float* dout, unsigned long long* dclk, int iters) {
  auto fn = k<MODE, LDS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {           // ~10 ms each: power management settles
    CK(hipEventRecord(e0));
    for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 131072, 0, din, dout, iters, dclk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[512];
  CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
  const double flop = 20.0 * 256 * 8 * (double)iters * 8 * 2.0 * 32 * 32 * ((MODE >= 2) ? 32 : 16);
  printf("%-28s %8.1f T/s   clock %.3f GHz (block 0)\n", name, flop / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] * 10.0));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  std::vector<uint32_t> hz(16384 * 4, 0), hr(16384 * 4), hs(16384 * 4);
  uint32_t st = 777;
  for (auto& v : hr) { st = st * 1664525u + 1013904223u; uint32_t x = st; st = st * 1664525u + 1013904223u; v = (x & 0x3BFFu) | (x & 0x8000u) | (((st >> 7) & 0x3BFFu) << 16) | (st & 0x80000000u); }   // finite fp16 pairs, |x| < 2
  for (auto& v : hs) { st = st * 1664525u + 1013904223u; v = st & 0x03030303u; }                                                            // small ints
  u32x4* din; float* dout; unsigned long long* dclk;
  CK(hipMalloc(&din, hr.size() * 4));
  CK(hipMalloc(&dout, 256 * 512 * 4));
  CK(hipMalloc(&dclk, 512 * 8));
  std::vector<uint32_t> h8(16384 * 4);
  for (auto& v : h8) { st = st * 1664525u + 1013904223u; v = st & 0xBFBFBFBFu; }            // e4m3 bytes, |x| < 2, no NaN: fp8 rows read this fill as "random"
  const char* fills[3] = {"zeros", "random", "small"};
  for (int fill = 0; fill < 3; ++fill) {
    CK(hipMemcpy(din, fill == 0 ? hz.data() : fill == 1 ? (argc > 2 ? h8.data() : hr.data()) : hs.data(), hr.size() * 4, hipMemcpyHostToDevice));
    printf("-- operand fill: %s\n", fills[fill]);
    run<0, false>("f16 32x32x16", din, dout, dclk, iters);
    run<1, false>("f16 16x16x32", din, dout, dclk, iters);
    run<0, true>("f16 32x32x16 + ds_read", din, dout, dclk, iters);
    run<1, true>("f16 16x16x32 + ds_read", din, dout, dclk, iters);
    run<2, false>("i8  32x32x32", din, dout, dclk, iters);
    run<3, false>("i8  16x16x64", din, dout, dclk, iters);
    run<2, true>("i8  32x32x32 + ds_read", din, dout, dclk, iters);
    run8<0>("fp8 16x16x32", din, dout, dclk, iters);
    run8<1>("fp8 16x16x128 scaled", din, dout, dclk, iters / 2);
    run8<2>("fp8 32x32x64 scaled", din, dout, dclk, iters / 2);
  }
  return 0;
}
