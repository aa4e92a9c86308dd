// conv_ppw.hip: the 8-wave ping-pong weight-gradient kernel (256 x 256 tiles of dw, 64-pixel K-tiles)
#pragma once
#include <hip/hip_runtime.h>

struct PpwArgs {
  const void* x;
  const void* dy;
  float* dw;
  int M;  // N*OH*OW
  int N, IH, IW, Cin, Cout, KH, KW, stride, pad_h, pad_w, OH, OW;
  int nCoB, nKB;
  int pix_per_split;
};
bool maai_wgrad_pp_supported(const PpwArgs& a);
int maai_wgrad_pp_launch(PpwArgs a, hipStream_t st, int target);
