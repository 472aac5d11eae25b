// member table: exact-product GEMV with the caller's RMSNorm folded into the activation staging (WQAA_EPI_RMSNORM_INPUT; plain
// stores, or the gate / up pair behind it), W 2-bit integer x A fp16
#include "wqaa_gemvx_kernel.h"
namespace wqaa {
gemvx_fn pick_gemvx_norm2(int layout, int mode, int mb, int rd) {
  return layout == LAYOUT_LOP3 ? pick_gemvx_norm_mode<2, LAYOUT_LOP3>(mode, mb, rd) : pick_gemvx_norm_mode<2, LAYOUT_PLAIN>(mode, mb, rd);
}
}  // namespace wqaa
