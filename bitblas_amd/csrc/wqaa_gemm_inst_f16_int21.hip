// member table: W int2/uint2/int1/uint1 x A fp16
#include "wqaa_gemm_kernel.h"
namespace wqaa {
gemm_fn pick_gemm_f16_int21(int kind, int layout, int mode, int mf) {
  if (kind == DK_INT2) return layout == LAYOUT_LOP3 ? pick_modes<DK_INT2, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT2, LAYOUT_PLAIN>(mode, mf);
  if (kind == DK_INT1) return layout == LAYOUT_LOP3 ? pick_modes<DK_INT1, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT1, LAYOUT_PLAIN>(mode, mf);
  return nullptr;
}
}  // namespace wqaa
