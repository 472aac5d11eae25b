// member table: the ping-pong members (wqaa_gemm_pp_kernel.h), 256 x 256 and 128 x 256 tiles - 4-bit weights x fp16,
// 2-bit weights x int8; dense fp8
#include "wqaa_gemm_pp_kernel.h"
namespace wqaa {

template <int KIND, int LAYOUT, int AT, int MODE, int BM>
using PPMember = PPPolicy<KIND, LAYOUT, AT, MODE, 0, BM == 128 ? 5 : 3, 0, BM>;

template <int KIND, int LAYOUT, int BM>
static gemm_fn pp_modes_f16(int mode) {
  switch (mode) {
    case MD_NONE: return wq_gemm_pp_kernel<PPMember<KIND, LAYOUT, AT_F16, MD_NONE, BM>>;
    case MD_S: return wq_gemm_pp_kernel<PPMember<KIND, LAYOUT, AT_F16, MD_S, BM>>;
    case MD_ZO: if constexpr (KIND == DK_INT4) return wq_gemm_pp_kernel<PPMember<KIND, LAYOUT, AT_F16, MD_ZO, BM>>; else return nullptr;
    case MD_ZR: if constexpr (KIND == DK_INT4) return wq_gemm_pp_kernel<PPMember<KIND, LAYOUT, AT_F16, MD_ZR, BM>>; else return nullptr;
    case MD_ZQ: if constexpr (KIND == DK_INT4) return wq_gemm_pp_kernel<PPMember<KIND, LAYOUT, AT_F16, MD_ZQ, BM>>; else return nullptr;
  }
  return nullptr;
}

template <int BM>
static gemm_fn pp_modes_bf16(int mode) {
  constexpr int R = BM == 128 ? 5 : 3;
  switch (mode) {
    case MD_NONE: return wq_gemm_pp_kernel<PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, R, 0, BM>>;
    case MD_S: return wq_gemm_pp_kernel<PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, R, 0, BM>>;
    case MD_ZO: return wq_gemm_pp_kernel<PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_ZO, FL_BF16, R, 0, BM>>;
    case MD_ZR: return wq_gemm_pp_kernel<PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_ZR, FL_BF16, R, 0, BM>>;
    case MD_ZQ: return wq_gemm_pp_kernel<PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_ZQ, FL_BF16, R, 0, BM>>;
  }
  return nullptr;
}

template <int BM>
static gemm_fn pick_pp_bm(int kind, int layout, int at, int mode, int* lds_bytes) {
  gemm_fn fn = nullptr;
  if (at == AT_F16) {
    if (kind == DK_INT4) fn = layout == LAYOUT_LOP3 ? pp_modes_f16<DK_INT4, LAYOUT_LOP3, BM>(mode) : pp_modes_f16<DK_INT4, LAYOUT_PLAIN, BM>(mode);
    else if (kind == DK_LUT4) fn = pp_modes_f16<DK_LUT4, LAYOUT_PLAIN, BM>(mode);
    if (fn) *lds_bytes = mode == MD_NONE ? PPMember<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_NONE, BM>::LDS_BYTES : PPMember<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_S, BM>::LDS_BYTES;
  } else if (at == AT_I8 && kind == DK_INT2 && mode == MD_NONE) {
    fn = layout == LAYOUT_LOP3 ? wq_gemm_pp_kernel<PPMember<DK_INT2, LAYOUT_LOP3, AT_I8, MD_NONE, BM>>
                               : wq_gemm_pp_kernel<PPMember<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, BM>>;
    *lds_bytes = PPMember<DK_INT2, LAYOUT_PLAIN, AT_I8, MD_NONE, BM>::LDS_BYTES;
  }
  return fn;
}

// nullptr: no ping-pong member for this combination (the caller falls back to wq_gemm_kernel)
gemm_fn pick_gemm_pp(int kind, int layout, int at, int mode, int flags, int bm, int bn, int* lds_bytes) {
  gemm_fn fn = nullptr;
  if (bn == 128) {                                      // the 128 x 128 tile: dense fp8 only
    if (!(bm == 128 && at == AT_F8 && mode == MD_NONE && (kind == DK_E4M3 || kind == DK_E5M2) && (flags & ~FL_ABF8) == 0)) return nullptr;
    const bool wb = kind == DK_E5M2, ab = (flags & FL_ABF8) != 0;
    *lds_bytes = PP8SPolicy<0, 0>::LDS_BYTES;
    return !wb ? (!ab ? wq_gemm_pp8s_kernel<PP8SPolicy<0, 0>> : wq_gemm_pp8s_kernel<PP8SPolicy<0, 1>>)
               : (!ab ? wq_gemm_pp8s_kernel<PP8SPolicy<1, 0>> : wq_gemm_pp8s_kernel<PP8SPolicy<1, 1>>);
  }
  if (bn != 256 || (bm != 256 && bm != 128)) return nullptr;
  // round 5: the dense 256 x 256 tile on a 2 x 4 wave grid (wq_gemm_pp8w_kernel).  WQAA_GEMM_TUNE=pp8_wide=0: the 1 x 8 grid (plan time)
  static const bool kWideDefault = true;
  int wide_knob = kWideDefault ? 1 : 0;
  (void)gemm_knob("pp8_wide", &wide_knob);
  const bool wide = bm == 256 && wide_knob != 0;
  if (wide && mode == MD_NONE) {
    if (at == AT_F8 && (kind == DK_E4M3 || kind == DK_E5M2) && (flags & ~FL_ABF8) == 0) {
      const bool wb = kind == DK_E5M2, ab = (flags & FL_ABF8) != 0;
      *lds_bytes = PP8WPolicy<0, 0>::LDS_BYTES;
      return !wb ? (!ab ? wq_gemm_pp8w_kernel<PP8WPolicy<0, 0>> : wq_gemm_pp8w_kernel<PP8WPolicy<0, 1>>)
                 : (!ab ? wq_gemm_pp8w_kernel<PP8WPolicy<1, 0>> : wq_gemm_pp8w_kernel<PP8WPolicy<1, 1>>);
    }
    if (at == AT_I8 && kind == DK_NATIVE && flags == 0) {
      *lds_bytes = PP8WPolicy<4, 4>::LDS_BYTES;
      return wq_gemm_pp8w_kernel<PP8WPolicy<4, 4>>;
    }
    if (at == AT_F16 && kind == DK_NATIVE && (flags & ~(int)FL_BF16) == 0) {
      *lds_bytes = PP8WPolicy<2, 2>::LDS_BYTES;
      return (flags & FL_BF16) ? wq_gemm_pp8w_kernel<PP8WPolicy<3, 3>> : wq_gemm_pp8w_kernel<PP8WPolicy<2, 2>>;
    }
  }
  if (at == AT_F8 && mode == MD_NONE && (kind == DK_E4M3 || kind == DK_E5M2) && (flags & ~FL_ABF8) == 0) {   // dense fp8 x fp8, all four pairings
    const bool wb = kind == DK_E5M2, ab = (flags & FL_ABF8) != 0;
    if (bm == 256) {
      fn = !wb ? (!ab ? wq_gemm_pp8_kernel<PP8Policy<0, 0>> : wq_gemm_pp8_kernel<PP8Policy<0, 1>>)
               : (!ab ? wq_gemm_pp8_kernel<PP8Policy<1, 0>> : wq_gemm_pp8_kernel<PP8Policy<1, 1>>);
      *lds_bytes = PP8Policy<0, 0>::LDS_BYTES;
    } else {
      fn = !wb ? (!ab ? wq_gemm_pp8_kernel<PP8Policy<0, 0, 0, 128>> : wq_gemm_pp8_kernel<PP8Policy<0, 1, 0, 128>>)
               : (!ab ? wq_gemm_pp8_kernel<PP8Policy<1, 0, 0, 128>> : wq_gemm_pp8_kernel<PP8Policy<1, 1, 0, 128>>);
      *lds_bytes = PP8Policy<0, 0, 0, 128>::LDS_BYTES;
    }
    return fn;
  }
  if (at == AT_I8 && kind == DK_NATIVE && mode == MD_NONE && flags == 0) {                      // dense int8 x int8 -> int32
    if (bm == 256) {
      fn = wq_gemm_pp8_kernel<PP8Policy<4, 4>>;
      *lds_bytes = PP8Policy<4, 4>::LDS_BYTES;
    } else {
      fn = wq_gemm_pp8_kernel<PP8Policy<4, 4, 0, 128>>;
      *lds_bytes = PP8Policy<4, 4, 0, 128>::LDS_BYTES;
    }
    return fn;
  }
  if (at == AT_F16 && kind == DK_NATIVE && mode == MD_NONE && (flags & ~(int)FL_BF16) == 0) {   // dense float16 / bfloat16 x the same type
    const bool bf = (flags & FL_BF16) != 0;
    if (bm == 256) {
      fn = bf ? wq_gemm_pp8_kernel<PP8Policy<3, 3>> : wq_gemm_pp8_kernel<PP8Policy<2, 2>>;
      *lds_bytes = PP8Policy<2, 2>::LDS_BYTES;
    } else {
      fn = bf ? wq_gemm_pp8_kernel<PP8Policy<3, 3, 0, 128>> : wq_gemm_pp8_kernel<PP8Policy<2, 2, 0, 128>>;
      *lds_bytes = PP8Policy<2, 2, 0, 128>::LDS_BYTES;
    }
    return fn;
  }
  if (flags == (int)FL_BF16 && at == AT_F16 && kind == DK_LUT4 && layout == LAYOUT_PLAIN && (mode == MD_NONE || mode == MD_S)) {   // bfloat16 x nf4 / fp4
    if (mode == MD_NONE) {
      fn = bm == 256 ? wq_gemm_pp_kernel<PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 3, 0, 256>>
                     : wq_gemm_pp_kernel<PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 5, 0, 128>>;
      *lds_bytes = bm == 256 ? PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 3, 0, 256>::LDS_BYTES
                             : PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 5, 0, 128>::LDS_BYTES;
    } else {
      fn = bm == 256 ? wq_gemm_pp_kernel<PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 3, 0, 256>>
                     : wq_gemm_pp_kernel<PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 5, 0, 128>>;
      *lds_bytes = bm == 256 ? PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 3, 0, 256>::LDS_BYTES
                             : PPPolicy<DK_LUT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 5, 0, 128>::LDS_BYTES;
    }
    return fn;
  }
  if (flags == (int)FL_BF16 && at == AT_F16 && kind == DK_INT4 && layout == LAYOUT_PLAIN) {      // bfloat16 activations x 4-bit integer weights
    fn = bm == 256 ? pp_modes_bf16<256>(mode) : pp_modes_bf16<128>(mode);
    if (fn) *lds_bytes = bm == 256 ? PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 3, 0, 256>::LDS_BYTES
                                   : PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_S, FL_BF16, 5, 0, 128>::LDS_BYTES;
    if (fn && mode == MD_NONE) *lds_bytes = bm == 256 ? PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 3, 0, 256>::LDS_BYTES
                                                      : PPPolicy<DK_INT4, LAYOUT_PLAIN, AT_F16, MD_NONE, FL_BF16, 5, 0, 128>::LDS_BYTES;
    return fn;
  }
  if (flags != 0) return nullptr;                       // strict e4m3, other bfloat16 formats: wq_gemm_kernel
  fn = bm == 256 ? pick_pp_bm<256>(kind, layout, at, mode, lds_bytes) : pick_pp_bm<128>(kind, layout, at, mode, lds_bytes);
  return fn;
}

}  // namespace wqaa
