// store_hazard_lab.hip - how many wait states a vector-memory store of 128 bits needs before its data registers may be overwritten
// (gfx950), by cache policy and by what fills the gap.  Round 5: an inline-assembly `global_store_dwordx4 ... sc0 sc1` whose first two
// data registers were re-used three SCALAR instructions later stored garbage in lanes 12-15 of every 16 (csrc/wqaa_gemm_mid_kernel.h);
// the compiler's own hazard recognizer puts `s_nop 1` there for stores it knows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_hazard_lab tools/store_hazard_lab.hip && tools/store_hazard_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// POLICY 0 plain, 1 nt, 2 sc0 sc1;  FILL 0: GAP x s_nop 0, 1: GAP x independent s_mov (SALU), 2: GAP x independent v_mov (VALU).
// One asm block per store: data built in v[40:43], stored, GAP fillers, then v40 AND v43 overwritten (nothing else can land in the gap).
#define LAB_ASM(POL, FILLINSN)                                                                                          \
  asm volatile("v_mov_b32 v40, %1\n\tv_xor_b32 v41, 0x11111111, %1\n\tv_xor_b32 v42, 0x22222222, %1\n\tv_xor_b32 v43, 0x33333333, %1\n\t" \
               "s_nop 4\n\t"                                                                                             \
               "global_store_dwordx4 %0, v[40:43], off" POL "\n\t"                                                        \
               ".rept %3\n\t" FILLINSN "\n\t.endr\n\t"                                                                  \
               "v_mov_b32 v40, %2\n\tv_mov_b32 v43, %2"                                                                  \
               ::"v"(dst), "v"(gid), "v"(junk), "n"(GAP) : "memory", "v40", "v41", "v42", "v43", "v44", "s40")
template <int POLICY, int FILL, int GAP>
__global__ void __launch_bounds__(256) k(u32x4* out, int reps) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  for (int r = 0; r < reps; ++r) {
    u32x4* dst = out + (size_t)r * gridDim.x * 256 + gid;
    const unsigned junk = 0xDEAD0000u + r;
    if (POLICY == 0) {
      if (FILL == 0) LAB_ASM("", "s_nop 0");
      else if (FILL == 1) LAB_ASM("", "s_mov_b32 s40, 0x1234");
      else LAB_ASM("", "v_mov_b32 v44, 0x1234");
    } else if (POLICY == 1) {
      if (FILL == 0) LAB_ASM(" nt", "s_nop 0");
      else if (FILL == 1) LAB_ASM(" nt", "s_mov_b32 s40, 0x1234");
      else LAB_ASM(" nt", "v_mov_b32 v44, 0x1234");
    } else {
      if (FILL == 0) LAB_ASM(" sc0 sc1", "s_nop 0");
      else if (FILL == 1) LAB_ASM(" sc0 sc1", "s_mov_b32 s40, 0x1234");
      else LAB_ASM(" sc0 sc1", "v_mov_b32 v44, 0x1234");
    }
  }
}

typedef void (*kfn)(u32x4*, int);
template <int POLICY, int FILL>
static void sweep(u32x4* d, std::vector<unsigned>& h, int grid, int reps) {
  const char* pol[] = {"plain", "nt", "sc0 sc1"};
  const char* fil[] = {"s_nop", "SALU", "VALU"};
  kfn fns[] = {k<POLICY, FILL, 0>, k<POLICY, FILL, 1>, k<POLICY, FILL, 2>, k<POLICY, FILL, 3>, k<POLICY, FILL, 4>, k<POLICY, FILL, 6>, k<POLICY, FILL, 8>, k<POLICY, FILL, 12>};
  const int gaps[] = {0, 1, 2, 3, 4, 6, 8, 12};
  printf("%-8s gap filled with %-6s:", pol[POLICY], fil[FILL]);
  for (int g = 0; g < 8; ++g) {
    CK(hipMemset(d, 0, h.size() * 4));
    hipLaunchKernelGGL(fns[g], dim3(grid), dim3(256), 0, 0, d, reps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    unsigned lanes = 0;
    for (size_t i = 0; i < h.size() / 4; ++i) {
      const unsigned gid = (unsigned)(i % ((size_t)grid * 256));
      if (h[4 * i] != gid) { ++bad; lanes |= 1u << ((gid & 63) >> 2); }
      if (h[4 * i + 1] != (gid ^ 0x11111111u) || h[4 * i + 2] != (gid ^ 0x22222222u) || h[4 * i + 3] != (gid ^ 0x33333333u)) ++bad;
    }
    printf("  gap %2d: %7zu bad (lane quads %04x)", gaps[g], bad, lanes);
  }
  printf("\n");
}

int main() {
  const int grid = 1024, reps = 64;
  std::vector<unsigned> h((size_t)grid * 256 * reps * 4);
  u32x4* d;
  CK(hipMalloc(&d, h.size() * 4));
  sweep<0, 0>(d, h, grid, reps); sweep<0, 1>(d, h, grid, reps); sweep<0, 2>(d, h, grid, reps);
  sweep<1, 0>(d, h, grid, reps); sweep<1, 1>(d, h, grid, reps); sweep<1, 2>(d, h, grid, reps);
  sweep<2, 0>(d, h, grid, reps); sweep<2, 1>(d, h, grid, reps); sweep<2, 2>(d, h, grid, reps);
  return 0;
}
