// BatchNorm-backward folded through a channel-expanding pointwise convolution (engine.py, "_FOLD"): the small dense algebra
// on the [Cout, Cin] / [Cin, Cin] side.  For y = x W^T followed by training-mode BatchNorm (conv3 + bn3 of a bottleneck,
// resnet.py:118-119) the backward dy = k1*g - k2 - k3*y enters the weight and data gradients only through
//     G1 = g^T x  [Cout, Cin]    Gram = x^T x  [Cin, Cin]    sx = colsum(x)  [Cin]        (pixel reductions: MFMA launches)
// and the kernels here turn those into what the big launches need:
//     fold_s2:       S2[c] = sum_k W[c,k] G1[c,k] - mean[c] S1[c]                          (BatchNorm-backward's second sum)
//     fold_dw:       dW[c,k] = k1[c] G1[c,k] - k2[c] sx[k] - k3[c] sum_j W[c,j] Gram[j,k]  (the weight gradient)
//     fold_dgrad_w:  Wf[k][c] = bf16(k1[c] W[c,k])                  data-gradient form [Cin][Cout] of the scaled weights
//                    Tn[k][j] = bf16(-sum_c W[c,j] k3[c] W[c,k])    [Cin][Cin]: dx -= x (W^T diag(k3) W)
//                    cn[k]    = -sum_c k2[c] W[c,k] - comp[k]       [Cin]:      dx -= k2 W; comp: the pixel mean of what rounding
//                                                                   Wf and Tn to bf16 adds to dx[:, k] (fold_wf_kernel)
//                    dg[k]    = -(W^T diag(k3) W)[k,k]              [Cin] fp32: T's diagonal, kept OUT of the bf16 matrix (fold_t_kernel)
// W is the bf16 copy the forward MFMAs multiplied by (kernel layout [Cout][Cin]).  All of it is a few hundred MFLOP per unit:
// plain fp32 loops, no tiling (measured: < 0.15 ms per unit at the largest shape, 2048 x 512).
#include "common.h"
#include "maai_internal.h"

__global__ __launch_bounds__(256) void fold_s2_kernel(const bf16_t* __restrict__ w, const float* __restrict__ g1, const double* __restrict__ s1,
                                                      const float* __restrict__ mean, double* __restrict__ s2, int Cout, int Cin) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= Cout) return;
  double acc = 0.0;
  for (int k = lane; k < Cin; k += 64) acc += (double)bf16_to_f32(w[(long long)c * Cin + k]) * (double)g1[(long long)c * Cin + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) s2[c] = acc - (double)mean[c] * s1[c];
}

// dW[c][k]: block = 256 consecutive k of one c (Cin >= 64: blocks of min(Cin, 256) threads)
__global__ __launch_bounds__(256) void fold_dw_kernel(const bf16_t* __restrict__ w, const float* __restrict__ g1, const float* __restrict__ gram,
                                                      const double* __restrict__ sx, const float* __restrict__ k1, const float* __restrict__ k2,
                                                      const float* __restrict__ k3, float* __restrict__ dw, int Cout, int Cin) {
  extern __shared__ float wrow[];   // W[c][:] as fp32
  const int c = blockIdx.y;
  for (int j = threadIdx.x; j < Cin; j += blockDim.x) wrow[j] = bf16_to_f32(w[(long long)c * Cin + j]);
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Cin) return;
  float acc = 0.f;
  for (int j = 0; j < Cin; ++j) acc = fmaf(wrow[j], gram[(long long)j * Cin + k], acc);
  dw[(long long)c * Cin + k] = k1[c] * g1[(long long)c * Cin + k] - k2[c] * (float)sx[k] - k3[c] * acc;
}

// Tn[k][j] = bf16(-sum_c W[c][j] k3[c] W[c][k]): block = one k (row of Tn), threads stride over j; k3 * W[:, k] staged in LDS.
// The DIAGONAL of T = W^T diag(k3) W is a sum of squares — sqrt(Cout) times larger than the off-diagonal entries — and the term
// it multiplies, x[p][k], is what the mask of output channel k (x_k > 0) and the unit below's second BatchNorm-backward sum
// (dx * (y2 - mean2), y2 -> x_k monotone) are built from: its bf16 rounding error would be the one coherent error of the folded
// data gradient.  With ``dg`` it never enters the bf16 matrix: Tn[k][k] = 0 and dg[k] = -T[k][k] goes out in fp32 for the
// epilogue of the data-gradient launch (dx[p][k] += dg[k] * x[p][k], x_k recomputed from the unit below's raw output).
// ct[k] = sum_{j} sx[j] * (Tn[k][j] - exact) / N: the pixel mean of what rounding the REST of Tn adds to dx[:, k] (fold_wf_kernel).
__global__ __launch_bounds__(256) void fold_t_kernel(const bf16_t* __restrict__ w, const float* __restrict__ k3, const double* __restrict__ sx,
                                                     double inv_n, bf16_t* __restrict__ tn, int tn_pitch, float* __restrict__ dg,
                                                     float* __restrict__ ct, int Cout, int Cin) {
  extern __shared__ float wk[];   // k3[c] * W[c][k]
  __shared__ float red[256];
  const int k = blockIdx.x;
  for (int c = threadIdx.x; c < Cout; c += 256) wk[c] = k3[c] * bf16_to_f32(w[(long long)c * Cin + k]);
  __syncthreads();
  float bias = 0.f;
  for (int j = threadIdx.x; j < Cin; j += 256) {
    float acc = 0.f;
    for (int c = 0; c < Cout; ++c) acc = fmaf(bf16_to_f32(w[(long long)c * Cin + j]), wk[c], acc);
    if (dg && j == k) {
      dg[k] = -acc;
      tn[(long long)k * tn_pitch + j] = f32_to_bf16(0.f);
      continue;
    }
    const bf16_t r = f32_to_bf16(-acc);
    tn[(long long)k * tn_pitch + j] = r;
    bias = fmaf((float)(sx[j] * inv_n), bf16_to_f32(r) + acc, bias);
  }
  red[threadIdx.x] = bias;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ct[k] = red[0];
}

// Wf[k][c] = bf16(k1[c] W[c][k]); cn[k] = -sum_c k2[c] W[c][k] - (rounding compensation): block = one k, threads over c.
// Rounding the folded weights to bf16 perturbs dx[p][k] by g[p] . dWf[k] + x[p] . dTn[k] — tiny per element, but COHERENT over
// the pixels (the same weight error meets the non-zero means of g and of the post-ReLU x), which is exactly what the
// BatchNorm-backward sums of the unit below add up.  Its pixel mean — (s1 . dWf[k] + sx . dTn[k]) / N (fold_t_kernel's ct[k]) — is
// known here and is taken out of the constant: the rounding error of the folded weights is then zero-mean over the pixels.
__global__ __launch_bounds__(256) void fold_wf_kernel(const bf16_t* __restrict__ w, const float* __restrict__ k1, const float* __restrict__ k2,
                                                      const double* __restrict__ s1, const float* __restrict__ ct, double inv_n,
                                                      bf16_t* __restrict__ wf, int wf_pitch, float* __restrict__ cn, int Cout, int Cin) {
  __shared__ float red[256], red2[256];
  const int k = blockIdx.x;
  float part = 0.f, bias = 0.f;
  for (int c = threadIdx.x; c < Cout; c += 256) {
    const float v = bf16_to_f32(w[(long long)c * Cin + k]);
    const float exact = k1[c] * v;
    const bf16_t r = f32_to_bf16(exact);
    wf[(long long)k * wf_pitch + c] = r;
    part = fmaf(k2[c], v, part);
    bias = fmaf((float)s1[c], bf16_to_f32(r) - exact, bias);
  }
  red[threadIdx.x] = part;
  red2[threadIdx.x] = bias;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      red2[threadIdx.x] += red2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) cn[k] = -red[0] - (float)((double)red2[0] * inv_n) - ct[k];
}

// Forward BatchNorm statistics of y = x W^T WITHOUT computing y: sum_p y[p][c] = W[c] . sx,  sum_p y[p][c]^2 = W[c] Gram W[c]^T
// (fp64 on the fp64 sums of a deterministic Gram pass, maai_gram_partials): sums[c] | sums[Cout + c] in the layout
// maai_bn_finalize takes.  One wave per output channel.
__global__ __launch_bounds__(256) void fold_stats_kernel(const bf16_t* __restrict__ w, const double* __restrict__ gram, const double* __restrict__ sx,
                                                         double* __restrict__ sums, int Cout, int Cin) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= Cout) return;
  const bf16_t* wr = w + (long long)c * Cin;
  double s = 0.0, q = 0.0;
  for (int k = lane; k < Cin; k += 64) {
    const double wk = (double)bf16_to_f32(wr[k]);
    s += wk * sx[k];
    double t = 0.0;
    for (int j = 0; j < Cin; ++j) t += (double)bf16_to_f32(wr[j]) * gram[(long long)j * Cin + k];
    q += wk * t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) {
    sums[c] = s;
    sums[Cout + c] = q;
  }
}

extern "C" int maai_fold_stats(const void* w, const double* gram, const double* sx, double* sums, int Cout, int Cin, void* stream) {
  MAAI_CHECK_ARG(w && gram && sx && sums && Cout > 0 && Cin > 0, "fold_stats: bad arguments");
  hipLaunchKernelGGL(fold_stats_kernel, dim3((Cout + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const bf16_t*)w, gram, sx, sums,
                     Cout, Cin);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_fold_s2(const void* w, const float* g1, const double* s1, const float* mean, double* s2, int Cout, int Cin, void* stream) {
  MAAI_CHECK_ARG(w && g1 && s1 && mean && s2 && Cout > 0 && Cin > 0, "fold_s2: bad arguments");
  hipLaunchKernelGGL(fold_s2_kernel, dim3((Cout + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const bf16_t*)w, g1, s1, mean, s2, Cout, Cin);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_fold_dw(const void* w, const float* g1, const float* gram, const double* sx, const float* k1, const float* k2,
                            const float* k3, float* dw, int Cout, int Cin, void* stream) {
  MAAI_CHECK_ARG(w && g1 && gram && sx && k1 && k2 && k3 && dw && Cout > 0 && Cin > 0 && Cin <= 8192, "fold_dw: bad arguments");
  const int nt = Cin < 256 ? ((Cin + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(fold_dw_kernel, dim3((Cin + nt - 1) / nt, Cout), dim3(nt), Cin * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                     (const bf16_t*)w, g1, gram, sx, k1, k2, k3, dw, Cout, Cin);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_fold_dgrad_w(const void* w, const float* k1, const float* k2, const float* k3, const double* s1, const double* sx,
                                 double count, void* wf, int wf_pitch, void* tn, int tn_pitch, float* cn, float* dg, float* scratch, int Cout,
                                 int Cin, void* stream) {
  MAAI_CHECK_ARG(w && k1 && k2 && k3 && s1 && sx && wf && tn && cn && scratch && count > 0 && Cout > 0 && Cin > 0 && Cout <= 8192 &&
                     wf_pitch >= Cout && tn_pitch >= Cin,
                 "fold_dgrad_w: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fold_t_kernel, dim3(Cin), dim3(256), Cout * sizeof(float), st, (const bf16_t*)w, k3, sx, 1.0 / count, (bf16_t*)tn, tn_pitch,
                     dg, scratch, Cout, Cin);
  hipLaunchKernelGGL(fold_wf_kernel, dim3(Cin), dim3(256), 0, st, (const bf16_t*)w, k1, k2, s1, scratch, 1.0 / count, (bf16_t*)wf, wf_pitch, cn, Cout, Cin);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
