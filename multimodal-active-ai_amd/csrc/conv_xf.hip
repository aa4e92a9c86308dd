// Forward convolutions whose INPUT is a raw (pre-BatchNorm) convolution output: the normalisation (+ ReLU) of the
// producing unit — resnet.py:101-109 `out = relu(bn1(conv1(x)))` feeding conv2, `relu(bn2(..))` feeding conv3 — is
// applied to the staged A operand in LDS (conv_igemm.h, XF), so `out` is never written to or read from HBM; with
// two raw tensors (XF == 2) it is the residual join `relu(bn3(y3) + identity)` of resnet.py:126-133 that the next
// block's first convolution forms on the way in, handing the joined activation back once for the shortcut.
// This file only holds the instantiations: the tile is chosen by conv_fwd.hip (ConvSel), so that the statistics
// slab has the rows maai_conv2d_stats_rows_fused() promises.
#include "conv_igemm.h"

template <typename T, int BM, int BN, int NSTAGE, int XF>
static int xf_rows(const ConvArgs& a, bool pw, hipStream_t st) {
  if constexpr (XF == 2) {
    return launch_conv_p<T, BM, BN, NSTAGE, 0, true, false, false, 2>(a, st);
  } else {
    return pw ? launch_conv_p<T, BM, BN, NSTAGE, 0, true, false, false, 1>(a, st)
              : launch_conv_p<T, BM, BN, NSTAGE, 0, false, false, false, 1>(a, st);
  }
}

template <typename T, int XF>
static int xf_128(const ConvArgs& a, const ConvSel& s, hipStream_t st) {
  if (s.bn == 128) return s.nstage == 2 ? xf_rows<T, 128, 128, 2, XF>(a, s.pw, st) : xf_rows<T, 128, 128, 3, XF>(a, s.pw, st);
  return s.nstage == 2 ? xf_rows<T, 128, 64, 2, XF>(a, s.pw, st) : xf_rows<T, 128, 64, 3, XF>(a, s.pw, st);
}

template <int XF>
static int xf_dispatch(const ConvArgs& a, const ConvSel& s, hipStream_t st) {
  if (s.dtype == MAAI_F32) {
    if (s.bm != 128 || s.halo || s.bn > 128) {
      maai_set_error("conv2d_igemm: fp32 normalise-on-load launches use the 128-row tile");
      return MAAI_ERR_UNSUPPORTED;
    }
    return xf_128<float, XF>(a, s, st);
  }
  if (s.halo) {
    if constexpr (XF == 1) {
      return s.bn == 128 ? launch_conv_p<bf16_t, 256, 128, 3, 0, false, true, false, 1>(a, st)
                         : launch_conv_p<bf16_t, 256, 64, 3, 0, false, true, false, 1>(a, st);
    } else {
      maai_set_error("conv2d_igemm: the two-tensor join on load is for pointwise layers");
      return MAAI_ERR_UNSUPPORTED;
    }
  }
  if (s.bn == 256) return xf_rows<bf16_t, 128, 256, 3, XF>(a, s.pw, st);
  if (s.bm == 256) return s.bn == 128 ? xf_rows<bf16_t, 256, 128, 3, XF>(a, s.pw, st) : xf_rows<bf16_t, 256, 64, 3, XF>(a, s.pw, st);
  return xf_128<bf16_t, XF>(a, s, st);
}

int maai_conv_xf_launch(const ConvArgs& a, const ConvSel& s, hipStream_t st) {
  if (a.xb) {
    if (!s.pw) {
      maai_set_error("conv2d_igemm: the two-tensor join on load is for pointwise layers");
      return MAAI_ERR_UNSUPPORTED;
    }
    return xf_dispatch<2>(a, s, st);
  }
  return xf_dispatch<1>(a, s, st);
}
