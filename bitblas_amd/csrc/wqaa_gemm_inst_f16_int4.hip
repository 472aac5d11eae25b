// member table: W int4/uint4 x A fp16 (both checkpoint layouts, all dequant modes)
#include "wqaa_gemm_kernel.h"
namespace wqaa {
gemm_fn pick_gemm_f16_int4(int layout, int mode, int mf) {
  return layout == LAYOUT_LOP3 ? pick_modes<DK_INT4, LAYOUT_LOP3>(mode, mf) : pick_modes<DK_INT4, LAYOUT_PLAIN>(mode, mf);
}
}  // namespace wqaa
