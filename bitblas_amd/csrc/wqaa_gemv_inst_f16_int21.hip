// member table: W int2/uint2/int1/uint1 x A fp16 GEMV (both checkpoint layouts, all dequant modes)
#include "wqaa_gemv_kernel.h"
namespace wqaa {
gemv_fn pick_gemv_f16_int4(int layout, int mode, int mb);
gemv_fn pick_gemv_f16_int(int kind, int layout, int mode, int mb) {
  switch (kind) {
    case DK_INT4: return pick_gemv_f16_int4(layout, mode, mb);
    case DK_INT2: return layout == LAYOUT_LOP3 ? pick_mode_f16<DK_INT2, LAYOUT_LOP3>(mode, mb) : pick_mode_f16<DK_INT2, LAYOUT_PLAIN>(mode, mb);
    case DK_INT1: return layout == LAYOUT_LOP3 ? pick_mode_f16<DK_INT1, LAYOUT_LOP3>(mode, mb) : pick_mode_f16<DK_INT1, LAYOUT_PLAIN>(mode, mb);
  }
  return nullptr;
}
}  // namespace wqaa
