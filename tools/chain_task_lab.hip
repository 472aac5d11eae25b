// chain_task_lab.hip - what the consumer side of csrc/wqaa_chain_kernel.h (chain_task: the exact-product decode / dots of
// lane chunks read back from LDS) sustains per CU by the number of consumer waves: shader cycles per 1 KiB unit (one weight
// row x one lane chunk), no weight stream, no waits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ibitblas_amd/csrc -o tools/chain_task_lab tools/chain_task_lab.hip
#include "wqaa_chain_kernel.h"
#include <cstdio>
#include <vector>
using namespace wqaa;
using P = GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ROWS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_task(int nc, int iters, unsigned long long* out, unsigned short* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 40 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x00010001u * (i & 7);
  __syncthreads();
  ChainTaskCtx X;
  X.nc = nc; X.cpr = nc * 64; X.kg = nc * 16; X.gq_shift = 2; X.N = 1 << 20; X.n0 = 0; X.gq_magic = 0; X.flip = 0;
  X.a_off = 0; X.sa_off = 24 * 1024; X.ring_off = 40 * 1024 + (wave % 4) * 24 * 1024; X.ring_units = 24;
  X.sc_rel[0] = X.sc_rel[1] = 28 * 1024; X.z_rel[0] = X.z_rel[1] = 28 * 1024;
  X.zint = 8.f; X.has_bias = 0; X.has_res = 0; X.stash_off = 0; X.bias[0] = X.bias[1] = nullptr; X.residual = nullptr;
  X.C = sink; X.gran = nullptr; X.tag = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int rpos = 0;
  for (int it = 0; it < iters; ++it) {
    chain_task<P, ROWS>(smem, X, (blockIdx.x * WAVES + wave) * 4 + (it & 3), rpos, lane);
    rpos += ROWS * nc;
    while (rpos >= 24) rpos -= 24;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int ROWS, int WAVES>
void run(int nc, unsigned long long* dout, unsigned short* sink) {
  const int iters = 64, G = 256;
  CK(hipFuncSetAttribute((const void*)k_task<ROWS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_task<ROWS, WAVES>), dim3(G), dim3(64 * WAVES), 136 * 1024, 0, nc, iters, dout, sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(G * WAVES);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double per_wave_unit = sum / h.size() / (iters * ROWS * nc);
  printf("rows/task %d  lane chunks %d  consumer waves/CU %2d : %7.1f cycles per unit per wave, %6.1f per unit per CU  (%.1f GB/s per CU at 2.35 GHz)\n", ROWS, nc, WAVES,
         per_wave_unit, per_wave_unit / WAVES, 1024.0 / (per_wave_unit / WAVES) * 2.35);
}

int main() {
  unsigned long long* dout;
  unsigned short* sink;
  CK(hipMalloc(&dout, 256 * 16 * 8));
  CK(hipMalloc(&sink, 8 << 20));
  for (int nc : {2, 6}) {
    run<4, 4>(nc, dout, sink); run<4, 8>(nc, dout, sink); run<4, 12>(nc, dout, sink); run<4, 16>(nc, dout, sink);
    run<2, 4>(nc, dout, sink); run<2, 8>(nc, dout, sink); run<2, 12>(nc, dout, sink); run<2, 16>(nc, dout, sink);
  }
  return 0;
}
