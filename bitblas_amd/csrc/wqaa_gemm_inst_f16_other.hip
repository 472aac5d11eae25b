// member table: W int8/uint8, nf4/fp4 (LUT), e4m3, e5m2, fp16 x A fp16
#include "wqaa_gemm_kernel.h"
namespace wqaa {
gemm_fn pick_gemm_f16_other(int kind, int mode, int flags, int mf) {
  switch (kind) {
    case DK_INT8: return pick_modes<DK_INT8, LAYOUT_PLAIN>(mode, mf);
    case DK_LUT4: return pick_modes_fp<DK_LUT4, 0>(mode, mf);
    case DK_E4M3: return (flags & FL_STRICT) ? pick_modes_fp<DK_E4M3, FL_STRICT>(mode, mf) : pick_modes_fp<DK_E4M3, 0>(mode, mf);
    case DK_E5M2: return pick_modes_fp<DK_E5M2, 0>(mode, mf);
    case DK_NATIVE: return mode == MD_NONE ? pick_mf<DK_NATIVE, LAYOUT_PLAIN, AT_F16, MD_NONE, 0>(mf) : nullptr;
  }
  return nullptr;
}

}  // namespace wqaa
