// wqaa_gemm.hip - MFMA GEMM family (placeholder while the GEMV slice is brought up)
#include "wqaa_common.h"
namespace wqaa {
int gemm_plan(const wqaa_matmul_desc&, int, wqaa_plan*) {
  set_error(WQAA_ERR_UNSUPPORTED, "gemm: not built yet");
  return WQAA_ERR_UNSUPPORTED;
}
int gemm_launch(const wqaa_matmul_desc&, const void*, const void*, const void*, const void*, const void*,
                const void*, void*, int, hipStream_t, hipEvent_t, hipEvent_t) {
  set_error(WQAA_ERR_UNSUPPORTED, "gemm: not built yet");
  return WQAA_ERR_UNSUPPORTED;
}
void gemm_init() {}
}  // namespace wqaa
