// member table: exact-product GEMV, W 4-bit integer x A fp16 (both checkpoint layouts, all dequant modes)
#include "wqaa_gemvx_kernel.h"
namespace wqaa {
gemvx_fn pick_gemvx_int4(int layout, int mode, int mb, int rd) {
  return layout == LAYOUT_LOP3 ? pick_gemvx_mode<4, LAYOUT_LOP3>(mode, mb, rd) : pick_gemvx_mode<4, LAYOUT_PLAIN>(mode, mb, rd);
}
}  // namespace wqaa
