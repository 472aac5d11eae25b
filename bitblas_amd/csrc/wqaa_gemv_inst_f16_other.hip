// member table: 8-bit integer, nf4 / fp4, fp8 and native weights x A fp16 GEMV; dense fp8 x fp8
#include "wqaa_gemv_kernel.h"
namespace wqaa {
gemv_fn pick_gemv_f16_other(int kind, int mode, int flags, int mb) {
  if (flags & FL_A8) {  // dense fp8 x fp8: both operands widened to fp16, fp32 accumulate
    if (mode != MD_NONE) return nullptr;
    if (kind == DK_E4M3) return pick_mb<DK_E4M3, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_A8>(mb);
    if (kind == DK_E5M2) return pick_mb<DK_E5M2, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_A8>(mb);
    return nullptr;
  }
  switch (kind) {
    case DK_INT8: return pick_mode_f16<DK_INT8, LAYOUT_PLAIN>(mode, mb);
    case DK_LUT4: return pick_mode_fp<DK_LUT4, 0>(mode, mb);
    case DK_E4M3: return (flags & FL_STRICT) ? pick_mode_fp<DK_E4M3, FL_STRICT>(mode, mb) : pick_mode_fp<DK_E4M3, 0>(mode, mb);
    case DK_E5M2: return pick_mode_fp<DK_E5M2, 0>(mode, mb);
    case DK_NATIVE: return mode == MD_NONE ? pick_mb<DK_NATIVE, LAYOUT_PLAIN, AT_F16, MD_NONE, 0>(mb) : nullptr;
  }
  return nullptr;
}
}  // namespace wqaa
