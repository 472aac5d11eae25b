// member table: the ping-pong 256 x 256 members (wqaa_gemm_pp_kernel.h) - 4-bit weights x fp16, 2-bit weights x int8
#include "wqaa_gemm_pp_kernel.h"
namespace wqaa {

template <int KIND, int LAYOUT>
static gemm_fn pp_modes_f16(int mode) {
  switch (mode) {
    case MD_NONE: return wq_gemm_pp_kernel<PPPolicy<KIND, LAYOUT, AT_F16, MD_NONE, 0>>;
    case MD_S: return wq_gemm_pp_kernel<PPPolicy<KIND, LAYOUT, AT_F16, MD_S, 0>>;
    case MD_ZO: if constexpr (KIND == DK_INT4) return wq_gemm_pp_kernel<PPPolicy<KIND, LAYOUT, AT_F16, MD_ZO, 0>>; else return nullptr;
    case MD_ZR: if constexpr (KIND == DK_INT4) return wq_gemm_pp_kernel<PPPolicy<KIND, LAYOUT, AT_F16, MD_ZR, 0>>; else return nullptr;
  }
  return nullptr;
}

// nullptr: no ping-pong member for this combination (the caller falls back to wq_gemm_kernel)
gemm_fn pick_gemm_pp(int kind, int layout, int at, int mode, int flags, int* lds_bytes) {
  gemm_fn fn = nullptr;
  if (at == AT_F8 && mode == MD_NONE && (kind == DK_E4M3 || kind == DK_E5M2) && (flags & ~FL_ABF8) == 0) {   // dense fp8 x fp8, all four pairings
    const bool wb = kind == DK_E5M2, ab = (flags & FL_ABF8) != 0;
    fn = !wb ? (!ab ? wq_gemm_pp8_kernel<PP8Policy<0, 0>> : wq_gemm_pp8_kernel<PP8Policy<0, 1>>)
             : (!ab ? wq_gemm_pp8_kernel<PP8Policy<1, 0>> : wq_gemm_pp8_kernel<PP8Policy<1, 1>>);
    *lds_bytes = PP8Policy<0, 0>::LDS_BYTES;
    return fn;
  }
  if (flags != 0) return nullptr;                       // bfloat16 / strict e4m3: wq_gemm_kernel
  if (at == AT_F16) {
    if (kind == DK_INT4) fn = layout == LAYOUT_LOP3 ? pp_modes_f16<DK_INT4, LAYOUT_LOP3>(mode) : pp_modes_f16<DK_INT4, LAYOUT_PLAIN>(mode);
    else if (kind == DK_LUT4) fn = pp_modes_f16<DK_LUT4, LAYOUT_PLAIN>(mode);
    if (fn) *lds_bytes = mode == MD_NONE ? PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_NONE, 0>::LDS_BYTES : PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_S, 0>::LDS_BYTES;
  } else if (at == AT_I8 && kind == DK_INT2 && mode == MD_NONE) {
    fn = layout == LAYOUT_LOP3 ? wq_gemm_pp_kernel<PPPolicy<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, 0>>
                               : wq_gemm_pp_kernel<PPPolicy<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>>;
    *lds_bytes = PPPolicy<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, 0>::LDS_BYTES;
  }
  return fn;
}

}  // namespace wqaa
