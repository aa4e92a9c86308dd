// Internal interface between conv_fwd.hip (dispatch) and conv_pw.hip (direct-epilogue pointwise kernel).
#pragma once
#include <hip/hip_runtime.h>

struct PwArgs {
  const void* x;      // [M, Cin] bf16 (NHWC pixels)
  const void* w;      // [Cout, Cin] bf16
  void* y;            // [M, Cout] bf16
  float* stats;       // partial slab [M/128][2][Cout] (forward statistics or BN-backward sums)
  const void* mask;   // optional ReLU mask laid out like y
  long long M;
  int Cin, Cout;
  int accumulate;
  int nMB, nNB;
  int erelu;
  const float* ep0;
  const float* ep1;
  const float* ep2;
  const void* et;
};

int maai_pw_conv_launch(PwArgs a, int emode, hipStream_t st);
