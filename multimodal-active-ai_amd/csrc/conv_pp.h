// Internal interface between conv_fwd.hip (dispatch) and conv_pp.hip (persistent pipelined pointwise kernel).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

struct PpArgs {
  const bf16_t* x;   // [M, Cin]
  const bf16_t* w;   // [Cout, Cin]
  bf16_t* y;         // [M, Cout]
  float* stats;      // partial slab [M/128][2][Cout] or NULL
  long long M;
  int Cin, Cout;
  int nMB, nNB, ntiles;
};

int maai_pp_conv_launch(PpArgs a, hipStream_t st);
