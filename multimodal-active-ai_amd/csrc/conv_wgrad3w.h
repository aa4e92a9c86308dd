// conv_wgrad3w.hip: the eight-wave wide patch kernel for 3x3 weight gradients, pad 1, stride 1 or 2 (128 x 9 x 64 blocks of dw)
#pragma once
#include <hip/hip_runtime.h>

struct Wgrad3wArgs {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, Cin, Cout;   // H, W: the output plane
  int IH, IW, stride;       // the input plane (= H, W at stride 1); 1 or 2
  int tilesX, tilesY;
  int nCoB, nCiB;
  long long npatch, per_split;
};
bool maai_wgrad3w_supported(const Wgrad3wArgs& a, bool* by_rule);
int maai_wgrad3w_launch(Wgrad3wArgs a, hipStream_t st, int target);
