// (round 5: moved out of the product library - csrc/ - with the WQAA_GEMVX_ABL switch; kept as the source of the round-2 ablation numbers)
// lab members: ablations of the int4 / LOP3 / scale / M = 1 / R = 2 exact-product GEMV (WQAA_GEMVX_ABL=<bits>, tools only)
#include "wqaa_gemvx_kernel.h"
namespace wqaa {
gemvx_fn pick_gemvx_lab(int abl) {
  switch (abl) {
    case 1: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 1>>;
    case 2: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 2>>;
    case 3: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 3>>;
    case 4: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 4>>;
    case 5: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 5>>;
    case 7: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 7>>;
    case 16: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 16>>;
    case 32: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 32>>;
    case 48: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 48>>;
    case 8: return wq_gemvx_kernel<GemvxPolicy<4, LAYOUT_LOP3, MD_S, 1, 2, 2, 8>>;
  }
  return nullptr;
}
}  // namespace wqaa
