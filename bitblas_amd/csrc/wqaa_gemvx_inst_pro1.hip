// member table: exact-product GEMV with the caller's gated activation / residual add folded in (WQAA_EPI_GATED_INPUT,
// WQAA_EPI_ADD_RESIDUAL), W 1-bit integer x A fp16
#include "wqaa_gemvx_kernel.h"
namespace wqaa {
gemvx_fn pick_gemvx_pro1(int layout, int mode, int mb, int rd) {
  return layout == LAYOUT_LOP3 ? pick_gemvx_pro_mode<1, LAYOUT_LOP3>(mode, mb, rd) : pick_gemvx_pro_mode<1, LAYOUT_PLAIN>(mode, mb, rd);
}
}  // namespace wqaa
