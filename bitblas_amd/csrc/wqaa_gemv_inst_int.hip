// member table: int8 / packed int4 activations (dot4 path), incl. the BitNet members that quantise fp16 activations themselves
#include "wqaa_gemv_kernel.h"
namespace wqaa {
gemv_fn pick_gemv_int(int kind, int layout, int at, int flags, int mb) {
  if (at == AT_I8 && (flags & FL_AQ)) {   // BitNet layers: fp16 activations quantised in the kernel (sub-byte weights)
    switch (kind) {
      case DK_INT4: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT4, LAYOUT_LOP3, AT_I8, MD_NONE, FL_AQ>(mb) : pick_mb<DK_INT4, LAYOUT_PLAIN, AT_I8, MD_NONE, FL_AQ>(mb);
      case DK_INT2: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, FL_AQ>(mb) : pick_mb<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, FL_AQ>(mb);
      case DK_INT1: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT1, LAYOUT_LOP3, AT_I8, MD_NONE, FL_AQ>(mb) : pick_mb<DK_INT1, LAYOUT_PLAIN, AT_I8, MD_NONE, FL_AQ>(mb);
    }
    return nullptr;
  }
  if (at == AT_I4) {   // packed int4 activations: native int4 weights, or 2-bit weights in either layout
    if (kind == DK_INT4) return layout == LAYOUT_PLAIN ? pick_mb<DK_INT4, LAYOUT_PLAIN, AT_I4, MD_NONE, 0>(mb) : nullptr;
    if (kind == DK_INT2) return layout == LAYOUT_LOP3 ? pick_mb<DK_INT2, LAYOUT_LOP3, AT_I4, MD_NONE, 0>(mb) : pick_mb<DK_INT2, LAYOUT_PLAIN, AT_I4, MD_NONE, 0>(mb);
    return nullptr;
  }
  switch (kind) {
    case DK_INT4: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT4, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mb) : pick_mb<DK_INT4, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mb);
    case DK_INT2: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mb) : pick_mb<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mb);
    case DK_INT1: return layout == LAYOUT_LOP3 ? pick_mb<DK_INT1, LAYOUT_LOP3, AT_I8, MD_NONE, 0>(mb) : pick_mb<DK_INT1, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mb);
    case DK_NATIVE: return pick_mb<DK_NATIVE, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>(mb);
  }
  return nullptr;
}
}  // namespace wqaa
