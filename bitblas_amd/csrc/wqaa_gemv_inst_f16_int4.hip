// member table: W int4/uint4 x A fp16 GEMV (both checkpoint layouts, all dequant modes)
#include "wqaa_gemv_kernel.h"
namespace wqaa {
gemv_fn pick_gemv_f16_int4(int layout, int mode, int mb) {
  return layout == LAYOUT_LOP3 ? pick_mode_f16<DK_INT4, LAYOUT_LOP3>(mode, mb) : pick_mode_f16<DK_INT4, LAYOUT_PLAIN>(mode, mb);
}
}  // namespace wqaa
