// member table: the mid-M one-launch split-K member (wqaa_gemm_mid_kernel.h) - W int4 / uint4 x A float16, both checkpoint
// layouts, every dequant mode; 32- / 64- / 128-row M-tiles (mf 2 / 4 / 8) x slices of 4 / 8 k-steps (nkh 2 / 4: K = 4096 / 8192 - the shapes
// the selector takes it for; nkh 1, K = 2048, measured slower than the member it stood in for and is not instantiated)
#include "wqaa_gemm_mid_kernel.h"
namespace wqaa {

template <int LAYOUT, int MODE, int MF, int NKH>
static gemm_fn mid_member(int* lds) {
  using P = MidPolicy<DK_INT4, LAYOUT, MODE, MF, NKH>;
  if (lds) *lds = P::LDS_BYTES;
  return wq_gemm_mid_kernel<P>;
}
template <int LAYOUT, int MODE>
static gemm_fn mid_shape(int mf, int nkh, int* lds) {
  // (the slice has to fit the CU's LDS: 16 MF rows x 2 NKH k-steps x 256 B <= 128 KiB)
  switch (mf * 10 + nkh) {
    case 82: return mid_member<LAYOUT, MODE, 8, 2>(lds);
    case 44: return mid_member<LAYOUT, MODE, 4, 4>(lds);
    case 42: return mid_member<LAYOUT, MODE, 4, 2>(lds);
    case 24: return mid_member<LAYOUT, MODE, 2, 4>(lds);
    case 22: return mid_member<LAYOUT, MODE, 2, 2>(lds);
  }
  return nullptr;
}
template <int LAYOUT>
static gemm_fn mid_mode(int mode, int mf, int nkh, int* lds) {
  switch (mode) {
    case MD_NONE: return mid_shape<LAYOUT, MD_NONE>(mf, nkh, lds);
    case MD_S: return mid_shape<LAYOUT, MD_S>(mf, nkh, lds);
    case MD_ZO: return mid_shape<LAYOUT, MD_ZO>(mf, nkh, lds);
    case MD_ZR: return mid_shape<LAYOUT, MD_ZR>(mf, nkh, lds);
    case MD_ZQ: return mid_shape<LAYOUT, MD_ZQ>(mf, nkh, lds);
  }
  return nullptr;
}
gemm_fn pick_gemm_mid(int kind, int layout, int mode, int mf, int nkh, int* lds_bytes) {
  if (kind != DK_INT4) return nullptr;
  return layout == LAYOUT_LOP3 ? mid_mode<LAYOUT_LOP3>(mode, mf, nkh, lds_bytes) : mid_mode<LAYOUT_PLAIN>(mode, mf, nkh, lds_bytes);
}
}  // namespace wqaa
