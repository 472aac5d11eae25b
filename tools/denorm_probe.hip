// denorm_probe.hip - does v_dot2c_f32_f16 / v_dot2_f32_f16 / v_pk_mul_f16 take fp16 DENORMAL inputs at face value on gfx950
// (default kernel mode)?  A 4-bit field AND-ed out of a packed word IS an fp16 denormal q * 2^-24.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
  const uint32_t q = 0x000f0007u;                 // halves: lo = 7 * 2^-24, hi = 15 * 2^-24
  const uint32_t qh = 0x00f00070u;                // lo = 7 * 2^-20, hi = 15 * 2^-20
  const half2_t a = {(_Float16)0.5f, (_Float16)-0.25f};
  float acc = 0.f;
  asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(q), "v"(a));
  out[0] = acc;                                    // expect (7*0.5 - 15*0.25) * 2^-24 = -0.25 * 2^-24
  float acc2 = 0.f;
  asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(qh), "v"(a));
  out[1] = acc2;                                   // expect -0.25 * 2^-20
  out[2] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, q), a, 0.f, false);
  uint32_t pm;
  asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(pm) : "v"(q), "v"(0x64006400u));   // * 1024 -> 7*2^-14, 15*2^-14 (still denormal)
  out[3] = (float)__builtin_bit_cast(half2_t, pm)[0];
  out[4] = (float)__builtin_bit_cast(half2_t, pm)[1];
  // accumulate many: 1000 x (15*2^-24 * 1.0)
  float acc3 = 0.f;
  const half2_t one = {(_Float16)1.f, (_Float16)1.f};
  for (int i = 0; i < 1000; ++i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc3) : "v"(0x000f000fu), "v"(one));
  out[5] = acc3;                                   // expect 30000 * 2^-24
}
int main() {
  float* d; hipMalloc(&d, 64); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  const float u = 1.f / 16777216.f;
  printf("dot2c denorm : %g (expect %g)\n", h[0], -0.25f * u);
  printf("dot2  denorm : %g (expect %g)\n", h[1], -0.25f * u * 16);
  printf("fdot2 builtin: %g (expect %g)\n", h[2], -0.25f * u);
  printf("pk_mul denorm: %g %g (expect %g %g)\n", h[3], h[4], 7 * 1024 * u, 15 * 1024 * u);
  printf("1000 x dot2c : %g (expect %g)\n", h[5], 30000 * u);
  return 0;
}
