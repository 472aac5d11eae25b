// member table: bfloat16 activations (plain layout; every dequant mode for the integer formats)
#include "wqaa_gemv_kernel.h"
namespace wqaa {
gemv_fn pick_gemv_bf16(int kind, int mode, int mb) {
  const bool zmode = mode == MD_ZO || mode == MD_ZR || mode == MD_ZQ;
#define WQAA_BF_PICK(K) \
  (mode == MD_NONE ? pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mb) \
   : mode == MD_S  ? pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16>(mb)    \
   : mode == MD_ZO ? pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_ZO, FL_BF16>(mb)   \
   : mode == MD_ZR ? pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_ZR, FL_BF16>(mb)   \
                   : pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_ZQ, FL_BF16>(mb))
#define WQAA_BF_PICK_NZ(K) \
  (mode == MD_NONE ? pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mb) : pick_mb<K, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16>(mb))
  switch (kind) {
    case DK_INT4: return WQAA_BF_PICK(DK_INT4);
    case DK_INT2: return WQAA_BF_PICK(DK_INT2);
    case DK_INT1: return WQAA_BF_PICK(DK_INT1);
    case DK_INT8: return WQAA_BF_PICK(DK_INT8);
    case DK_LUT4: return zmode ? nullptr : WQAA_BF_PICK_NZ(DK_LUT4);      // table / fp8 formats never pair with zero points
    case DK_E4M3: return zmode ? nullptr : WQAA_BF_PICK_NZ(DK_E4M3);
    case DK_NATIVE: return mode == MD_NONE ? pick_mb<DK_NATIVE, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16>(mb) : nullptr;
  }
#undef WQAA_BF_PICK
#undef WQAA_BF_PICK_NZ
  return nullptr;
}
}  // namespace wqaa
