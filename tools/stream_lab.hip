// stream_lab.hip - lab (round 5): how fast does the chip stream a row-major 4-bit weight matrix as a function of the ORDER in which its
// waves ask for it?  The GEMV family reads such a matrix at 5+ TB/s, every MFMA decode form (one-launch, persistent, K-sliced, with
// register loads or LDS-DMA) at 2.5 - 3.9.  Here 256 workgroups x 8 waves only stream: LDS-DMA of 16 B per lane (1 KiB per instruction)
// into a 12-slot ring per wave, `s_waitcnt vmcnt(11)` before each new one (12 KiB in flight per wave, 96 KiB per CU), the data never
// used.  A pattern is the map (wave, instruction, lane) -> (row, byte):
//   gemv   wave = 2 rows at a time, whole K: instruction = 1 row x 1 KiB, rows alternate, chunks in order      (the GEMV family)
//   gdyn   the same, the row pairs handed out at run time (one atomic per pair, asked for one pair ahead)
//   xdl    workgroup = 16-row fragment, wave = an eighth of K: instruction = 4 rows x 256 B, blocks of 4        (one-launch decode forms)
//   ksl    wave = fragment x an eighth of K (the workgroup's), 4 rows x 256 B                                   (K-sliced form)
//   f1k    wave = fragment x an eighth of K, instruction = 1 row x 1 KiB (16 rows, then the next KiB)
//   r4k    wave = 4 rows, whole K: instruction = 1 row x 1 KiB, the 4 rows alternate
//   r16k   wave = 16 rows (a fragment), whole K: 1 row x 1 KiB, the 16 rows alternate
//   r16q   wave = 16 rows, whole K: 4 rows x 256 B, blocks of 4 (a fragment's whole row walked by ONE wave)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/stream_lab tools/stream_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { P_GEMV, P_XDL, P_KSL, P_F1K, P_R4K, P_R16K, P_R16Q, P_GDYN };

template <int PAT>
__global__ void __launch_bounds__(512) k_stream(const uint8_t* W, long row_bytes, int N, uint32_t* out, unsigned long long* stamps) {
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x, wg = blockIdx.x;
  const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (12 * 1024);
  const int nfrags = N / 16;
  const long eighth = row_bytes / 8;                       // (multiples of 256 B in every shape the harness runs)
  long n_instr = 0;
  // per pattern: number of instructions of this wave, and (row, byte) of lane for instruction i
  int frag0 = 0, nfr = 0;
  if (PAT == P_GEMV || PAT == P_R4K || PAT == P_R16K || PAT == P_R16Q) {
    constexpr int RW = PAT == P_GEMV ? 2 : PAT == P_R4K ? 4 : 16;       // rows a wave walks together
    const int groups = N / RW, gw = wg * 8 + wave, nw = nwg * 8;
    nfr = gw < groups ? (groups - 1 - gw) / nw + 1 : 0;                  // row groups of this wave: gw, gw + nw, ...
    frag0 = gw;
    n_instr = (long)nfr * RW * (row_bytes / 1024);
  } else if (PAT == P_XDL) {
    nfr = wg < nfrags ? (nfrags - 1 - wg) / nwg + 1 : 0;                 // fragments of this workgroup: wg, wg + nwg, ...
    frag0 = wg;
    n_instr = (long)nfr * 16 * (eighth / 1024.0) > 0 ? (long)nfr * (16 * eighth / 1024) : 0;
  } else {                                                               // ksl, f1k: workgroup = (slice, group), wave = fragment
    const int groups = nwg / 8, grp = wg >> 3;
    const int first = grp * 8 + wave, wps = groups * 8;
    nfr = first < nfrags ? (nfrags - 1 - first) / wps + 1 : 0;
    frag0 = first;
    n_instr = (long)nfr * (16 * eighth / 1024);
  }
  auto address = [&](long i) -> const uint8_t* {
    long row, byte;
    if (PAT == P_GEMV || PAT == P_R4K || PAT == P_R16K) {
      constexpr int RW = PAT == P_GEMV ? 2 : PAT == P_R4K ? 4 : 16;
      const long per = RW * (row_bytes / 1024);
      const long g = i / per, j = i % per;
      row = ((long)frag0 + g * (nwg * 8)) * RW + (j % RW);
      byte = (j / RW) * 1024 + lane * 16;
    } else if (PAT == P_R16Q) {
      const long per = 16 * (row_bytes / 1024);
      const long g = i / per, j = i % per;
      row = ((long)frag0 + g * (nwg * 8)) * 16 + (j & 3) * 4 + (lane >> 4);
      byte = (j >> 2) * 256 + (lane & 15) * 16;
    } else if (PAT == P_XDL) {
      const long per = 16 * eighth / 1024;                                // instructions per fragment and wave
      const long g = i / per, j = i % per;
      row = ((long)frag0 + g * nwg) * 16 + (j & 3) * 4 + (lane >> 4);
      byte = wave * eighth + (j >> 2) * 256 + (lane & 15) * 16;
    } else if (PAT == P_KSL) {
      const long per = 16 * eighth / 1024;
      const long g = i / per, j = i % per;
      row = ((long)frag0 + g * ((nwg / 8) * 8)) * 16 + (j & 3) * 4 + (lane >> 4);
      byte = (wg & 7) * eighth + (j >> 2) * 256 + (lane & 15) * 16;
    } else {                                                              // f1k (the eighth a multiple of 1 KiB)
      const long per = 16 * eighth / 1024;
      const long g = i / per, j = i % per;
      row = ((long)frag0 + g * ((nwg / 8) * 8)) * 16 + (j & 15);
      byte = (wg & 7) * eighth + (j >> 4) * 1024 + lane * 16;
    }
    return W + row * row_bytes + byte;
  };
  auto dma = [&](long i, int slot) __attribute__((always_inline)) {
    const uint8_t* p = address(i);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + (uint32_t)slot * 1024);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  };
  if constexpr (PAT == P_GDYN) {
    // the gemv order with the row pairs handed out at run time: one atomic per pair, asked for one pair ahead (its latency rides under the
    // pair in hand); `out` is this launch's counter (zero at launch)
    const int pairs = N / 2, per = (int)(2 * (row_bytes / 1024));
    unsigned nxt = __hip_atomic_fetch_add(out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int slot = 0, inflight = 0;
    while ((int)__builtin_amdgcn_readfirstlane(nxt) < pairs) {
      const int pr = __builtin_amdgcn_readfirstlane(nxt);
      nxt = __hip_atomic_fetch_add(out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = 0; j < per; ++j) {
        if (inflight == 12) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else ++inflight;
        const uint8_t* p = W + ((long)pr * 2 + (j & 1)) * row_bytes + (long)(j >> 1) * 1024 + lane * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring + (uint32_t)slot * 1024);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
        slot = slot == 11 ? 0 : slot + 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamps && lane == 0) { stamps[(wg * 8 + wave) * 2] = t_start; stamps[(wg * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
    return;
  }
  if (n_instr == 0) {
    if (stamps && (threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t_start; stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t_start; }
    return;
  }
  const long last = n_instr - 1;
#pragma unroll
  for (int s = 0; s < 12; ++s) dma(s < n_instr ? s : last, s);
  for (long i = 12; i < n_instr; i += 12) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      dma(i + s < n_instr ? i + s : last, s);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps && lane == 0) { stamps[(wg * 8 + wave) * 2] = t_start; stamps[(wg * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
  if (out && smem[threadIdx.x] == 0x5a && smem[threadIdx.x + 512] == 0xa5) out[wg * 512 + threadIdx.x] = 1;
}

template <int PAT>
static void run(const char* name, std::vector<uint8_t*>& W, long N, long row_bytes, uint32_t* out, hipStream_t st) {
  auto fn = k_stream<PAT>;
  const int lds = 8 * 12 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int launches = 32;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    if (PAT == P_GDYN) CK(hipMemsetAsync(out, 0, 64 * 4, st));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, st, W[l % W.size()], row_bytes, (int)N, PAT == P_GDYN ? out + l : out, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double us = best * 1000.0 / launches, bytes = (double)N * row_bytes;
  printf("  %-6s %8.2f us/launch  %6.2f TB/s\n", name, us, bytes / us / 1e6);
  // one more launch with per-wave stamps (100 MHz): when do the waves of each XCD (workgroup % 8) finish?
  unsigned long long* d; CK(hipMalloc(&d, 2048 * 16));
  if (PAT == P_GDYN) CK(hipMemsetAsync(out, 0, 64 * 4, st));
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, st, W[0], row_bytes, (int)N, out, d);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> h(4096);
  CK(hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < 2048; ++w) t0 = h[2 * w] < t0 ? h[2 * w] : t0;
  printf("         last wave of XCD 0..7 done at (us):");
  double all_max = 0, all_mean = 0;
  for (int x = 0; x < 8; ++x) {
    double mx = 0, mean = 0; int n = 0;
    for (int w = 0; w < 2048; ++w) if (((w / 8) & 7) == x) { const double e = (h[2 * w + 1] - t0) / 100.0; mx = e > mx ? e : mx; mean += e; ++n; }
    printf(" %5.1f (mean %5.1f)", mx, mean / n);
    all_max = mx > all_max ? mx : all_max; all_mean += mean / n / 8;
  }
  printf("   chip: mean %5.1f max %5.1f\n", all_mean, all_max);
  CK(hipFree(d));
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
  printf("N = %ld, K = %ld (rows of %ld B, %.1f MB), %d buffers; back-to-back launches (the launch gap is in the number)\n", N, K, row_bytes, wbytes / 1e6, nbuf);
  if (row_bytes % 1024 == 0) run<P_GEMV>("gemv", W, N, row_bytes, out, st);
  if (row_bytes % 1024 == 0) run<P_GDYN>("gdyn", W, N, row_bytes, out, st);
  if (argc > 3) return 0;
  if ((row_bytes / 8) % 256 == 0) run<P_XDL>("xdl", W, N, row_bytes, out, st);
  if ((row_bytes / 8) % 256 == 0) run<P_KSL>("ksl", W, N, row_bytes, out, st);
  if ((row_bytes / 8) % 1024 == 0) run<P_F1K>("f1k", W, N, row_bytes, out, st);
  if (row_bytes % 1024 == 0) run<P_R4K>("r4k", W, N, row_bytes, out, st);
  if (row_bytes % 1024 == 0) run<P_R16K>("r16k", W, N, row_bytes, out, st);
  if (row_bytes % 1024 == 0) run<P_R16Q>("r16q", W, N, row_bytes, out, st);
  return 0;
}
