// granule_lab.hip - lab: does the SHAPE of a wave's load instruction bound what a CU ingests?  (round 5, after the K-sliced decode form
// measured a 3.8 TB/s slope where the GEMV family streams 5.9.)  Every wave walks 16-row weight fragments of a row-major N x row_bytes
// matrix in blocks of 256 bytes per row (4 KiB per block: four 16-byte-per-lane loads, three blocks in flight, nothing else done with
// the data), and only the lane -> address map of an instruction differs:
//   shape 16: 16 rows x  64 B per instruction (lane = row + 16 * chunk: the MFMA operand order the decode members load in)
//   shape  4:  4 rows x 256 B per instruction
//   shape  1:  1 row x 1 KiB per instruction (block = 1 KiB per row: sixteen loads) - the GEMV family's order
// Same bytes, same number of instructions, same depth.  Prints TB/s over the whole matrix for register loads and for LDS-DMA.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/granule_lab tools/granule_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool DMA>
__global__ void __launch_bounds__(512) k_walk(const uint8_t* W, long row_bytes, int nfrags, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  constexpr int LPR = 64 / SHAPE;                       // lanes per row in one instruction
  constexpr int BLK = SHAPE == 1 ? 1024 : 256;          // bytes per row and block
  constexpr int NI = 16 / SHAPE * (BLK / (LPR * 16));   // instructions per block
  static_assert(NI * 1024 == 16 * BLK, "a block is 16 rows x BLK bytes");
  const int nblk = (int)(row_bytes / BLK);
  u32x4 acc = {0, 0, 0, 0};
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (3 * NI * 1024);
  for (int f = gw; f < nfrags; f += nw) {
    const uint8_t* fb = W + (long)f * 16 * row_bytes;
    // lane's row within an instruction's rows, and its byte offset within the block's row segment
    const int r_in = lane / LPR, c_in = (lane % LPR) * 16;
    auto issue = [&](int b, int slot, u32x4 (&v)[NI]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < NI; ++q) {
        const uint8_t* p;
        if (SHAPE == 16) p = fb + (long)(lane & 15) * row_bytes + (long)b * BLK + q * 64 + (lane >> 4) * 16;   // the MFMA operand order
        else p = fb + (long)(q * SHAPE + r_in) * row_bytes + (long)b * BLK + c_in;
        if (DMA) {
          const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(slot * NI * 1024 + q * 1024));
          uint32_t keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
        } else {
          asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[q]) : "v"(p) : "memory");
        }
      }
    };
    u32x4 v0[NI], v1[NI], v2[NI];
    auto use = [&](u32x4 (&v)[NI]) __attribute__((always_inline)) {
      if (!DMA) {
#pragma unroll
        for (int q = 0; q < NI; ++q) acc ^= v[q];
      }
    };
    issue(0, 0, v0);
    issue(1 < nblk ? 1 : 0, 1, v1);
    issue(2 < nblk ? 2 : 0, 2, v2);
    for (int b = 0; b < nblk; b += 3) {
      {
        if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]) : "n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        use(v0);
        issue(b + 3 < nblk ? b + 3 : 0, 0, v0);
      }
      {
        if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3]) : "n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        use(v1);
        issue(b + 4 < nblk ? b + 4 : 0, 1, v1);
      }
      {
        if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v2[0]), "+v"(v2[1]), "+v"(v2[2]), "+v"(v2[3]) : "n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        use(v2);
        issue(b + 5 < nblk ? b + 5 : 0, 2, v2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[gw] = acc[2] ^ acc[3];
}

template <int SHAPE, bool DMA>
static void run(const char* name, std::vector<uint8_t*>& W, long N, long row_bytes, uint32_t* out, hipStream_t st) {
  auto fn = k_walk<SHAPE, DMA>;
  const int lds = DMA ? 8 * 3 * (SHAPE == 1 ? 16 : 4) * 1024 : 0;
  if (lds > 160 * 1024) { printf("%-28s (LDS)\n", name); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int launches = 32, nfrags = (int)(N / 16);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, st, W[l % W.size()], row_bytes, nfrags, out);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double us = best * 1000.0 / launches, bytes = (double)N * row_bytes;
  printf("%-28s %8.2f us/launch  %6.2f TB/s\n", name, us, bytes / us / 1e6);
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 8192, K = argc > 2 ? atol(argv[2]) : 28672;
  const long row_bytes = K / 2;
  const size_t wbytes = (size_t)N * row_bytes;
  const int nbuf = (int)((768ull << 20) / wbytes) < 2 ? 2 : (int)((768ull << 20) / wbytes);
  std::vector<uint8_t*> W(nbuf);
  for (auto& p : W) { CK(hipMalloc(&p, wbytes)); CK(hipMemset(p, 1, wbytes)); }
  uint32_t* out; CK(hipMalloc(&out, 1 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  printf("N = %ld, K = %ld (4-bit rows of %ld B, %.1f MB), %d buffers, 256 workgroups x 8 waves, 3 blocks of 4 KiB (shape 1: 16 KiB) in flight per wave\n", N, K, row_bytes, wbytes / 1e6, nbuf);
  run<16, false>("16 rows x 64 B  registers", W, N, row_bytes, out, st);
  run<4, false>(" 4 rows x 256 B registers", W, N, row_bytes, out, st);
  if (row_bytes % 1024 == 0) run<1, false>(" 1 row x 1 KiB  registers", W, N, row_bytes, out, st);
  run<16, true>("16 rows x 64 B  LDS-DMA", W, N, row_bytes, out, st);
  run<4, true>(" 4 rows x 256 B LDS-DMA", W, N, row_bytes, out, st);
  return 0;
}
