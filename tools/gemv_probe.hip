// gemv_probe.hip - ablation vehicle for the int4 x fp16 M=1 GEMV (N=K=4096, g=128, LOP3 layout).
// Variants are template flags so each piece of the kernel can be priced on the GPU box:
//   F_COMPUTE decode+dot (else xor the loaded words), F_LDS stage A through LDS, F_STORE write C,
//   F_SCALE load scales.  Timed as one hipGraph of NBUF launches over rotating weight buffers.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/gemv_probe tools/gemv_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
enum { F_COMPUTE = 1, F_LDS = 2, F_STORE = 4, F_SCALE = 8, F_XCD = 16, F_FOLD = 32, F_ADIRECT = 64 };
__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// K = 4096 fixed: 2 lane chunks per row. R rows per wave, NWAVE waves per block.
template <int FLAGS, int R, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64) k_gemv(const uint8_t* __restrict__ B, const half_t* __restrict__ A,
                                                     const uint16_t* __restrict__ S, half_t* __restrict__ C, int N) {
  constexpr int K = 4096, NC = 2, KG = K / 128;
  __shared__ u32x4 a_lds[K / 8];   // natural order: LOP3 extraction pairs are (2i, 2i+1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x;
  if (FLAGS & F_XCD) { const int per = gridDim.x / 8; blk = (blockIdx.x % 8) * per + blockIdx.x / 8; }
  const int rg = blk * NWAVE + wave;
  u32x4 areg[2];
  if (FLAGS & F_LDS) {
#pragma unroll
    for (int j = 0; j < (K / 8 + NWAVE * 64 - 1) / (NWAVE * 64); ++j) {
      const int i = j * NWAVE * 64 + tid;
      areg[j] = reinterpret_cast<const u32x4*>(A)[i < K / 8 ? i : 0];
    }
  }
  u32x4 adir[NC][4];
  if (FLAGS & F_ADIRECT) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) adir[c][u] = reinterpret_cast<const u32x4*>(A)[(c * 64 + lane) * 4 + u];
  }
  u32x4 w[R][NC]; uint32_t sc[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = rg * R + r;
      w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(B + (long)n * (K / 2) + (c * 64 + lane) * 16));
      if (FLAGS & F_SCALE) sc[r][c] = S[n * KG + (c * 64 + lane) / 4];
    }
  if (FLAGS & F_LDS) {
#pragma unroll
    for (int j = 0; j < (K / 8 + NWAVE * 64 - 1) / (NWAVE * 64); ++j) {
      const int i = j * NWAVE * 64 + tid;
      if (i < K / 8) a_lds[i] = areg[j];
    }
    __syncthreads();
  }
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  if (FLAGS & F_COMPUTE) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u32x4 av = (FLAGS & F_LDS) ? a_lds[(c * 64 + lane) * 4 + u] : (FLAGS & F_ADIRECT) ? adir[c][u] : u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t x = w[r][c][u];
          const half2_t s2 = (FLAGS & F_SCALE) ? half2_t{__builtin_bit_cast(half_t, (uint16_t)sc[r][c]), __builtin_bit_cast(half_t, (uint16_t)sc[r][c])} : half2_t{(half_t)1, (half_t)1};
          float part = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int bit = 4 * i, b = bit & 7;
            const uint32_t src = bit >= 8 ? (x >> 8) : x;
            const uint32_t t = (src & ((0xFu << b) * 0x00010001u)) | ((uint32_t)((25 - b) << 10) * 0x00010001u);
            const half_t off = (half_t)((float)(1 << (10 - b)) + 8.0f);
            half2_t q = as_h2(t) - half2_t{off, off};
            if (FLAGS & F_FOLD) part = __builtin_amdgcn_fdot2(q, as_h2(av[i]), part, false);
            else { q = q * s2; acc[r] = __builtin_amdgcn_fdot2(q, as_h2(av[i]), acc[r], false); }
          }
          if (FLAGS & F_FOLD) acc[r] += part * (float)s2[0];
        }
      }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        uint32_t x = w[r][c][0] ^ w[r][c][1] ^ w[r][c][2] ^ w[r][c][3];
        if (FLAGS & F_SCALE) x ^= sc[r][c];
        acc[r] += __builtin_bit_cast(float, x & 0x3fffffffu);
      }
    if (FLAGS & F_LDS) acc[0] += __builtin_bit_cast(float, a_lds[tid][0] & 0x3fffffffu);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float tot = wave_sum(acc[r]);
    if (FLAGS & F_STORE) { if (lane == 0) C[rg * R + r] = (half_t)tot; }
    else if (tot == 1234.5f) C[0] = (half_t)tot;
  }
}


// pipelined variant: a wave owns row groups rg, rg + total_waves, ...; the loads of the NEXT row group
// are in flight while the current one is decoded (two register stages, loop unrolled by two so that
// every load is unconditional inside the steady state)
template <int R, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64) k_gemv_pipe(const uint8_t* __restrict__ B, const half_t* __restrict__ A,
                                                          const uint16_t* __restrict__ S, half_t* __restrict__ C, int N) {
  constexpr int K = 4096, NC = 2, KG = K / 128;
  __shared__ u32x4 a_lds[K / 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) { const int per = gridDim.x / 8; blk = (blockIdx.x % 8) * per + blockIdx.x / 8; }
  const int total = gridDim.x * NWAVE;
  const int n_rg = N / R;
  int rg = blk * NWAVE + wave;
  u32x4 areg[2];
#pragma unroll
  for (int j = 0; j < (K / 8 + NWAVE * 64 - 1) / (NWAVE * 64); ++j) {
    const int i = j * NWAVE * 64 + tid;
    areg[j] = reinterpret_cast<const u32x4*>(A)[i < K / 8 ? i : 0];
  }
  struct St { u32x4 w[R][NC]; uint32_t sc[R][NC]; };
  auto issue = [&](St& st, int g) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int n = g * R + r;
        st.w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(B + (long)n * (K / 2) + (c * 64 + lane) * 16));
        st.sc[r][c] = S[n * KG + (c * 64 + lane) / 4];
      }
  };
  auto consume = [&](const St& st, int g) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u32x4 av = a_lds[(c * 64 + lane) * 4 + u];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t x = st.w[r][c][u];
          const half_t sh = __builtin_bit_cast(half_t, (uint16_t)st.sc[r][c]);
          const half2_t s2 = {sh, sh};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int bit = 4 * i, b = bit & 7;
            const uint32_t src = bit >= 8 ? (x >> 8) : x;
            const uint32_t t = (src & ((0xFu << b) * 0x00010001u)) | ((uint32_t)((25 - b) << 10) * 0x00010001u);
            const half_t off = (half_t)((float)(1 << (10 - b)) + 8.0f);
            half2_t q = as_h2(t) - half2_t{off, off};
            q = q * s2;
            acc[r] = __builtin_amdgcn_fdot2(q, as_h2(av[i]), acc[r], false);
          }
        }
      }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float tot = wave_sum(acc[r]);
      if (lane == 0) C[g * R + r] = (half_t)tot;
    }
  };
  St s0, s1;
  const bool work = rg < n_rg;
  issue(s0, work ? rg : n_rg - 1);
#pragma unroll
  for (int j = 0; j < (K / 8 + NWAVE * 64 - 1) / (NWAVE * 64); ++j) {
    const int i = j * NWAVE * 64 + tid;
    if (i < K / 8) a_lds[i] = areg[j];
  }
  __syncthreads();
  if (!work) return;
  while (true) {
    int nx = rg + total;
    if (nx >= n_rg) { consume(s0, rg); break; }
    issue(s1, nx);
    consume(s0, rg);
    rg = nx;
    nx = rg + total;
    if (nx >= n_rg) { consume(s1, rg); break; }
    issue(s0, nx);
    consume(s1, rg);
    rg = nx;
  }
}

template <class F>
static double time_graph(F launch_one, int nbuf, hipStream_t s) {
  for (int i = 0; i < nbuf; ++i) launch_one(i);
  CK(hipStreamSynchronize(s));
  hipGraph_t gr; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nbuf; ++i) launch_one(i);
  CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> v;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3f / nbuf);
  }
  std::sort(v.begin(), v.end());
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096, K = 4096, nbuf = argc > 1 ? 28 : 80;
  const size_t wbytes = (size_t)N * K / 2, sbytes = (size_t)N * (K / 128) * 2;
  std::vector<uint8_t> h(wbytes); srand(1); for (auto& b : h) b = (uint8_t)rand();
  std::vector<uint16_t> hs(sbytes / 2); for (auto& v : hs) v = 0x2000 | (rand() & 0x3ff);
  std::vector<uint16_t> ha(K); for (auto& v : ha) v = (rand() & 1 ? 0x8000 : 0) | 0x3000 | (rand() & 0x7ff);
  std::vector<uint8_t*> W(nbuf); std::vector<uint16_t*> S(nbuf);
  for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&W[i], wbytes)); CK(hipMemcpy(W[i], h.data(), wbytes, hipMemcpyHostToDevice));
                                   CK(hipMalloc(&S[i], sbytes)); CK(hipMemcpy(S[i], hs.data(), sbytes, hipMemcpyHostToDevice)); }
  half_t *A, *C; CK(hipMalloc(&A, K * 2)); CK(hipMalloc(&C, N * 2)); CK(hipMemcpy(A, ha.data(), K * 2, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  const double alg = wbytes + sbytes + K * 2 + N * 2;
#define RUN(FLAGS, R, NWAVE) { double us = time_graph([&](int i) { hipLaunchKernelGGL((k_gemv<FLAGS, R, NWAVE>), dim3(N / (R * NWAVE)), dim3(NWAVE * 64), 0, s, (const uint8_t*)W[i], (const half_t*)A, (const uint16_t*)S[i], C, N); }, nbuf, s); \
    printf("flags=%3d (%s%s%s%s%s%s%s) R=%d waves/block=%d grid=%4d : %6.3f us  -> %6.1f GB/s\n", FLAGS, (FLAGS & 1) ? "compute " : "", (FLAGS & 2) ? "lds " : "", (FLAGS & 4) ? "store " : "", (FLAGS & 8) ? "scale " : "", (FLAGS & 16) ? "xcd " : "", (FLAGS & 32) ? "fold " : "", (FLAGS & 64) ? "adirect " : "", R, NWAVE, N / (R * NWAVE), us, alg / us * 1e-3); }
  RUN(0, 2, 4) RUN(15 + 16, 2, 4) RUN(13 + 16 + 64, 2, 4) RUN(13 + 16 + 64, 1, 4) RUN(13 + 16 + 64, 2, 2) RUN(13 + 16 + 64, 4, 4) RUN(13 + 16 + 64, 2, 8)
#define RUNP(R, NWAVE, GRID) { double us = time_graph([&](int i) { hipLaunchKernelGGL((k_gemv_pipe<R, NWAVE>), dim3(GRID), dim3(NWAVE * 64), 0, s, (const uint8_t*)W[i], (const half_t*)A, (const uint16_t*)S[i], C, N); }, nbuf, s); \
    printf("pipelined R=%d waves/block=%d grid=%4d : %6.3f us  -> %6.1f GB/s\n", R, NWAVE, GRID, us, alg / us * 1e-3); }
  return 0;
}
