// prefetch_lab.hip - lab (round 6): what would a launch gain if its FIRST loads hit the L2 of its XCD?
// The GEMV family sits on the floor of what an independent launch streams (tools/stream_lab.hip: 8.4 MB in 4.5 - 5.1 us, 117 MB at
// 5.6 TB/s): launch boundary + the ramp until the first HBM bytes arrive + the drain.  A caller that knows the NEXT launch's weights
// (a decode step does) could let every wave of launch i touch the lines wave (workgroup, wave) of launch i + 1 asks for first - one
// `global_load_dword` per 128-byte line, 64 lines per instruction, same workgroup index = same XCD = same L2.  This lab prices it
// before anything is built: 256 workgroups x 8 waves stream a 4-bit matrix in the GEMV family's order (tools/stream_lab.hip `gemv`),
// launch i optionally touches the first P KiB per wave of buffer i + 1 at a chosen point of its own stream.
//   base     no touch
//   pfP@F    touch the first P KiB per wave of the next buffer after fraction F of the wave's own instructions
//   tlbP@F   one dword per wave only (the page walk, not the lines)
// Prints us / launch back to back, and from one stamped launch the mean time from the launch's first wave to a wave's first data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/prefetch_lab tools/prefetch_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// mode: 0 none, 1 lines, 2 one dword per wave
__global__ void __launch_bounds__(512) k_stream(const uint8_t* W, const uint8_t* Wnext, long row_bytes, int N, int mode, int pf_kib, int pf_at_256,
                                                uint32_t* out, unsigned long long* stamps) {
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x, wg = blockIdx.x;
  const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (12 * 1024);
  const int groups = N / 2, gw = wg * 8 + wave, nw = nwg * 8;
  const int nfr = gw < groups ? (groups - 1 - gw) / nw + 1 : 0;
  const int chunks = (int)(row_bytes >> 10), per = 2 * chunks;      // instructions (1 KiB each) per row pair: row 0 chunk 0, row 1 chunk 0, row 0 chunk 1, ...
  const int n_instr = nfr * per;
  if (n_instr == 0) {
    if (stamps && lane == 0) { stamps[gw * 3] = t_start; stamps[gw * 3 + 1] = 0; stamps[gw * 3 + 2] = t_start; }
    return;
  }
  uint32_t sink = 0, pv[4] = {0, 0, 0, 0};
  const int pf_i = (n_instr * pf_at_256) >> 8;                      // the instruction after which the touch goes out
  int g = 0, c = 0, r = 0, slot = 0;
  unsigned long long t_first = 0;
  for (int i = 0; i < n_instr; ++i) {
    if (i >= 12) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    if (i == 12) t_first = __builtin_amdgcn_s_memrealtime();
    const uint8_t* p = W + ((long)(gw + g * nw) * 2 + r) * row_bytes + c * 1024 + lane * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + (uint32_t)slot * 1024);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    slot = slot == 11 ? 0 : slot + 1;
    r ^= 1;
    if (!r && ++c == chunks) { c = 0; ++g; }
    if (mode && i == pf_i) {
      if (mode == 5) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      const uint8_t* nb = mode == 4 ? W : Wnext;
      if (mode != 2) {
#pragma unroll
        for (int b = 0; b < 32; b += 8) {                           // 8 KiB = 64 lines per instruction: lane -> (instruction b + lane / 8, line lane % 8)
          int ti = b + (lane >> 3);
          const bool on = ti < pf_kib;
          if (ti >= n_instr) ti = n_instr - 1;
          const int tg = ti / per, tj = ti % per;
          const uint8_t* q = nb + ((long)(gw + tg * nw) * 2 + (tj & 1)) * row_bytes + (tj >> 1) * 1024 + (lane & 7) * 128;
          // (the destination is a register the compiler never allocates: a load that lands later must not hit a live value)
          if (on) asm volatile("global_load_dword v100, %0, off" : : "v"(q) : "memory", "v100");
        }
      } else {
        const uint8_t* q = nb + ((long)gw * 2) * row_bytes;
        if (lane == 0) asm volatile("global_load_dword v100, %0, off" : : "v"(q) : "memory", "v100");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  sink = pv[0] ^ pv[1] ^ pv[2] ^ pv[3];
  if (stamps && lane == 0) { stamps[gw * 3] = t_start; stamps[gw * 3 + 1] = t_first; stamps[gw * 3 + 2] = __builtin_amdgcn_s_memrealtime(); }
  if (out && ((smem[threadIdx.x] == 0x5a && smem[threadIdx.x + 512] == 0xa5) || sink == 0x12345678u)) out[wg * 512 + threadIdx.x] = 1;
}

static void run(const char* name, std::vector<uint8_t*>& W, long N, long row_bytes, int mode, int pf_kib, int pf_at_256, uint32_t* out, hipStream_t st) {
  const int lds = 8 * 12 * 1024;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int launches = 48;
  const size_t nb = W.size();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int l = 0; l < launches; ++l)
      hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), lds, st, W[l % nb], W[(l + 1) % nb], row_bytes, (int)N, mode, pf_kib, pf_at_256, out, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double us = best * 1000.0 / launches, bytes = (double)N * row_bytes;
  // two launches, the second stamped: it runs behind a launch that has (or has not) touched its first lines
  unsigned long long* d; CK(hipMalloc(&d, 2048 * 24));
  double first_sum = 0, span_sum = 0; int reps = 0;
  for (int rep = 0; rep < 6; ++rep) {
    const int a = (2 * rep) % nb, b = (2 * rep + 1) % nb, c = (2 * rep + 2) % nb;
    hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), lds, st, W[a], W[b], row_bytes, (int)N, mode, pf_kib, pf_at_256, out, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), lds, st, W[b], W[c], row_bytes, (int)N, mode, pf_kib, pf_at_256, out, d);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(2048 * 3);
    CK(hipMemcpy(h.data(), d, 2048 * 24, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 2048; ++w) { t0 = h[3 * w] < t0 ? h[3 * w] : t0; t1 = h[3 * w + 2] > t1 ? h[3 * w + 2] : t1; }
    double f = 0; int n = 0;
    for (int w = 0; w < 2048; ++w) if (h[3 * w + 1]) { f += (h[3 * w + 1] - t0) / 100.0; ++n; }
    if (rep) { first_sum += f / (n ? n : 1); span_sum += (t1 - t0) / 100.0; ++reps; }
  }
  printf("  %-10s %7.2f us/launch  %5.2f TB/s   stamped launch: first data of a wave at %5.2f us (mean), first wave -> last wave done %5.2f us\n",
         name, us, bytes / us / 1e6, first_sum / reps, span_sum / reps);
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long N = argc > 1 ? atol(argv[1]) : 4096, K = argc > 2 ? atol(argv[2]) : 4096;
  const int only = argc > 3 ? atoi(argv[3]) : -1;
  const long row_bytes = K / 2;
  const size_t wbytes = (size_t)N * row_bytes;
  const int nbuf = (int)((1024ull << 20) / wbytes) < 3 ? 3 : (int)((1024ull << 20) / wbytes);
  std::vector<uint8_t*> W(nbuf);
  for (auto& p : W) { CK(hipMalloc(&p, wbytes)); CK(hipMemset(p, 1, wbytes)); }
  uint32_t* out; CK(hipMalloc(&out, 1 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("N = %ld, K = %ld (rows of %ld B, %.1f MB), %d buffers in rotation; 256 workgroups x 8 waves, 12 KiB in flight per wave\n", N, K, row_bytes, wbytes / 1e6, nbuf);
  if (row_bytes % 1024) { printf("  (rows must be multiples of 1 KiB)\n"); return 0; }
  for (int round = 0; round < 2; ++round) {
    if (only < 0 || only == 0) run("base", W, N, row_bytes, 0, 0, 0, out, st);
    if (only < 0 || only == 1) run("pf8@0", W, N, row_bytes, 1, 8, 0, out, st);
    if (only < 0 || only == 2) run("pf8@1/2", W, N, row_bytes, 1, 8, 128, out, st);
    if (only < 0 || only == 3) run("pf16@1/2", W, N, row_bytes, 1, 16, 128, out, st);
    if (only < 0 || only == 4) run("pf16@3/4", W, N, row_bytes, 1, 16, 192, out, st);
    if (only < 0 || only == 5) run("pf32@1/2", W, N, row_bytes, 1, 32, 128, out, st);
    if (only < 0 || only == 6) run("tlb@1/2", W, N, row_bytes, 2, 0, 128, out, st);
    if (only == 7) run("own8@1/2", W, N, row_bytes, 4, 8, 128, out, st);
    if (only == 8) run("nop-pf8@1/2", W, N, row_bytes, 5, 8, 128, out, st);
  }
  return 0;
}
