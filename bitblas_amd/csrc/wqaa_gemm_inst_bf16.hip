// member table: bfloat16 activations (test_general_matmul_bf16.py; support matrix README.md:63-70): plain-layout integer weights
// in every dequant mode, nf4 / fp4 / e4m3 with and without scale, dense bf16
#include "wqaa_gemm_kernel.h"
namespace wqaa {
template <int KIND>
static gemm_fn pick_bf_modes(int mode, int mf) {
  switch (mode) {
    case MD_NONE: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mf);
    case MD_S: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16>(mf);
    case MD_ZO: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_ZO, FL_BF16>(mf);
    case MD_ZR: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_ZR, FL_BF16>(mf);
    case MD_ZQ: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_ZQ, FL_BF16>(mf);
  }
  return nullptr;
}
template <int KIND>
static gemm_fn pick_bf_nozero(int mode, int mf) {
  switch (mode) {
    case MD_NONE: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mf);
    case MD_S: return pick_mf<KIND, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16>(mf);
  }
  return nullptr;
}
gemm_fn pick_gemm_bf16(int kind, int mode, int mf) {
  switch (kind) {
    case DK_INT4: return pick_bf_modes<DK_INT4>(mode, mf);
    case DK_INT2: return pick_bf_modes<DK_INT2>(mode, mf);
    case DK_INT8: return pick_bf_modes<DK_INT8>(mode, mf);
    case DK_INT1: return pick_bf_modes<DK_INT1>(mode, mf);
    case DK_LUT4: return pick_bf_nozero<DK_LUT4>(mode, mf);
    case DK_E4M3: return pick_bf_nozero<DK_E4M3>(mode, mf);
    case DK_NATIVE: return mode == MD_NONE ? pick_mf<DK_NATIVE, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mf) : nullptr;
  }
  return nullptr;
}
}  // namespace wqaa
