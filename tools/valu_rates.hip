// valu_rates.hip - issue cost of the instructions the GEMV decode is made of, on gfx950, measured with s_memtime:
// cycles per wave-instruction with W waves per SIMD (W = 1: latency-free issue interval of one wave; W = 8: what a
// saturated SIMD sustains).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { K_ANDOR = 0, K_PKADD, K_PKMUL, K_PKFMA, K_DOT2C, K_DOT2, K_FMA32, K_PERM, K_CVT, K_MFMA4, K_MFMA16x32, K_DOT2_MIX, K_MFMA4_MIX, K_MFMA16_MIX, K_DOT4, K_LSHR, K_NKINDS };
static const char* kNames[] = {"v_and_or_b32", "v_pk_add_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_dot2c_f32_f16", "v_dot2_f32_f16", "v_fma_f32", "v_perm_b32",
                               "v_cvt_f32_f16", "v_mfma_f32_4x4x4_16B_f16", "v_mfma_f32_16x16x32_f16", "mix: 13 valu + 4 dot2c (one word)",
                               "mix: 13 valu + 2 mfma4x4x4 (one word)", "mix: 13 valu + 1 mfma16x16x32 (one word)", "v_dot4_i32_i8", "v_lshrrev_b32"};

template <int KIND>
__global__ void __launch_bounds__(512) rate_kernel(uint64_t* out, int iters, uint32_t seed) {
  uint32_t a[8], b[8];
  float f[8];
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed * (i + 1) + threadIdx.x;
    b[i] = 0x3c003c00u + i;
    f[i] = 0.f;
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  uint32_t m = 0x64006400u;
  asm volatile("" : "+v"(m));
  uint32_t mk0 = 0xf000fu, mk1 = 0xf000f0u;
  asm volatile("" : "+s"(mk0), "+s"(mk1));
  uint64_t t0 = __builtin_readcyclecounter();
  t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_ANDOR) {
#define X(i) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m), "s"(mk0));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_LSHR) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(a[i]) : "v"(a[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_PKADD) {
#define X(i) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(m));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_DOT2C) {
#define X(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[i]) : "v"(a[i]), "v"(b[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_DOT2) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a[i]), "v"(b[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_DOT4) {
#define X(i) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(m));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_FMA32) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a[i]), "v"(b[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(m));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_CVT) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (KIND == K_MFMA4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          half4_t av = __builtin_bit_cast(half4_t, ((unsigned long)a[i] << 32) | b[i]);
          acc[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, av, acc[i], 0, 0, 0);
        }
    } else if constexpr (KIND == K_MFMA16x32) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          typedef uint32_t u4 __attribute__((ext_vector_type(4)));
          u4 v = {a[i], b[i], a[(i + 1) & 7], b[(i + 1) & 7]};
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, v), __builtin_bit_cast(half8_t, v), acc[i], 0, 0, 0);
        }
    } else if constexpr (KIND == K_DOT2_MIX || KIND == K_MFMA4_MIX || KIND == K_MFMA16_MIX) {
      // the instruction mix of one 32-bit weight word (8 int4) of the strict decode, 4 words per iteration = "32 units"
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint32_t x = a[w], x8, q0, q1, q2, q3;
        asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(q0) : "v"(x), "v"(m), "s"(mk0));
        asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(q1) : "v"(x), "v"(m), "s"(mk1));
        asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(x8) : "v"(x));
        asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(q2) : "v"(x8), "v"(m), "s"(mk0));
        asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(q3) : "v"(x8), "v"(m), "s"(mk1));
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(q0) : "v"(q0), "v"(b[0]));
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(q1) : "v"(q1), "v"(b[1]));
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(q2) : "v"(q2), "v"(b[0]));
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(q3) : "v"(q3), "v"(b[1]));
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(q0) : "v"(q0), "v"(b[2]));
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(q1) : "v"(q1), "v"(b[2]));
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(q2) : "v"(q2), "v"(b[2]));
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(q3) : "v"(q3), "v"(b[2]));
        if constexpr (KIND == K_DOT2_MIX) {
          asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[w]) : "v"(q0), "v"(b[4]));
          asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[w]) : "v"(q1), "v"(b[5]));
          asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[w]) : "v"(q2), "v"(b[6]));
          asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[w]) : "v"(q3), "v"(b[7]));
        } else if constexpr (KIND == K_MFMA4_MIX) {
          half4_t w0 = __builtin_bit_cast(half4_t, ((unsigned long)q1 << 32) | q0), w1 = __builtin_bit_cast(half4_t, ((unsigned long)q3 << 32) | q2);
          half4_t a0 = __builtin_bit_cast(half4_t, ((unsigned long)b[5] << 32) | b[4]), a1 = __builtin_bit_cast(half4_t, ((unsigned long)b[7] << 32) | b[6]);
          acc[w] = __builtin_amdgcn_mfma_f32_4x4x4f16(w0, a0, acc[w], 0, 0, 0);
          acc[w] = __builtin_amdgcn_mfma_f32_4x4x4f16(w1, a1, acc[w], 0, 0, 0);
        } else {
          typedef uint32_t u4 __attribute__((ext_vector_type(4)));
          u4 wv = {q0, q1, q2, q3}, av = {b[4], b[5], b[6], b[7]};
          acc[w] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wv), __builtin_bit_cast(half8_t, av), acc[w], 0, 0, 0);
        }
        a[w] = x + q0;   // keep the chain alive without adding an instruction the strict decode does not have... (1 add)
      }
    }
  }
  const uint64_t t1 = clock64();
  uint32_t sink = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) sink += a[i] + __builtin_bit_cast(uint32_t, f[i]) + __builtin_bit_cast(uint32_t, acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
  if (sink == 0x12345678u) out[0] = sink;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(uint64_t* d_out, int waves_per_simd) {
  const int iters = 2000;
  const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd);           // up to 8 waves per block
  const int blocks_per_cu = waves_per_simd > 2 ? waves_per_simd / 2 : 1;
  const int blocks = 256 * blocks_per_cu;
  const int nwaves = blocks * threads / 64;
  CK(hipMemset(d_out, 0, (1 + nwaves) * 8));
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 10, 3u);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 3u);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> h(1 + nwaves);
  CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin() + 1, h.end());
  const double med = (double)h[1 + nwaves / 2];
  const int per_iter = (KIND == K_DOT2_MIX || KIND == K_MFMA4_MIX || KIND == K_MFMA16_MIX) ? 4 : 32;   // units per iteration
  const double per_wave = med / ((double)iters * per_iter);
  printf("  %-44s W=%d  %7.2f ticks per %s per wave  -> %6.2f ticks of SIMD time per %s\n", kNames[KIND], waves_per_simd, per_wave,
         per_iter == 4 ? "word" : "instr", per_wave / waves_per_simd, per_iter == 4 ? "word" : "instr");
}

int main() {
  uint64_t* d_out;
  CK(hipMalloc(&d_out, (1 + 256 * 64) * 8));
  printf("clock64() ticks (s_memtime: constant 100 MHz? shader clock? - compare v_fma_f32 = 2 shader cycles at W=8)\n");
  for (int w : {1, 2, 4, 8}) {
    run<K_FMA32>(d_out, w); run<K_ANDOR>(d_out, w); run<K_LSHR>(d_out, w); run<K_PKADD>(d_out, w); run<K_PKMUL>(d_out, w); run<K_PKFMA>(d_out, w);
    run<K_DOT2C>(d_out, w); run<K_DOT2>(d_out, w); run<K_DOT4>(d_out, w); run<K_PERM>(d_out, w); run<K_CVT>(d_out, w);
    run<K_MFMA4>(d_out, w); run<K_MFMA16x32>(d_out, w);
    run<K_DOT2_MIX>(d_out, w); run<K_MFMA4_MIX>(d_out, w); run<K_MFMA16_MIX>(d_out, w);
    printf("\n");
  }
  return 0;
}
