// member table: int8 activations (W int4/int2/int1/int8, int32 accumulate) and dense fp8 x fp8
#include "wqaa_gemm_kernel.h"
namespace wqaa {
gemm_fn pick_gemm_i8_f8(int kind, int layout, int at, int flags, int mf) {
  if (at == AT_F8) {   // all four e4m3 / e5m2 pairings (general_matmul/__init__.py:33-51)
    const bool ab = (flags & FL_ABF8) != 0;
    if (kind == DK_E4M3) return ab ? pick_mf<DK_E4M3, LAYOUT_PLAIN, AT_F8, MD_NONE, FL_ABF8>(mf) : pick_mf<DK_E4M3, LAYOUT_PLAIN, AT_F8, MD_NONE, 0>(mf);
    if (kind == DK_E5M2) return ab ? pick_mf<DK_E5M2, LAYOUT_PLAIN, AT_F8, MD_NONE, FL_ABF8>(mf) : pick_mf<DK_E5M2, LAYOUT_PLAIN, AT_F8, MD_NONE, 0>(mf);
    return nullptr;
  }
  if (at == AT_I4) {   // packed int4 activations: native int4 weights, or 2-bit weights in either layout
    if (kind == DK_INT4) return layout == LAYOUT_PLAIN ? pick_mf<DK_INT4, LAYOUT_PLAIN, AT_I4, MD_NONE, 0>(mf) : nullptr;
    if (kind == DK_INT2) return layout == LAYOUT_LOP3 ? pick_mf<DK_INT2, LAYOUT_LOP3, AT_I4, MD_NONE, 0>(mf) : pick_mf<DK_INT2, LAYOUT_PLAIN, AT_I4, MD_NONE, 0>(mf);
    return nullptr;
  }
  switch (kind) {
    case DK_INT4: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT4, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT4, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_INT2: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_INT1: return layout == LAYOUT_LOP3 ? pick_mf<DK_INT1, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mf) : pick_mf<DK_INT1, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
    case DK_NATIVE: return pick_mf<DK_NATIVE, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mf);
  }
  return nullptr;
}
}  // namespace wqaa
