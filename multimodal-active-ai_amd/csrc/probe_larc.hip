// Two small consumers of the hot path's products:
//  * softmax cross-entropy of the linear probe (Representation_Evaluation.py:621-666: LogisticRegression logits ->
//    nn.CrossEntropyLoss, mean reduction, class-index targets) forward and backward, one wave per row;
//  * LARC (Model_Util.py:80-83, `--optimizer lars` = apex.parallel.LARC(Adam)): ||w|| and ||g|| of EVERY parameter in
//    one launch, and the per-tensor trust-ratio rescaling of the gradients in a second one — instead of four tiny
//    torch kernels per tensor (~640 launches per step, each shorter than its issue time).
#include "common.h"
#include "maai_internal.h"

#define ST(stream) reinterpret_cast<hipStream_t>(stream)

// ---------------------------------------------------------------------------
// softmax cross-entropy: logits [B][ld] (C <= ld valid columns), labels int64 [B]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                             float* __restrict__ loss, float* __restrict__ lse, int B, int C, int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* x = logits + (long long)row * ld;
  float m = -3.0e38f;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  const float l = m + __logf(s);
  if (lane == 0) {
    lse[row] = l;
    const long long y = labels[row];
    const float nll = (y >= 0 && y < C) ? l - x[y] : 0.f;   // out-of-range targets contribute nothing (checked on the host)
    atomicAdd(loss, nll / (float)B);
  }
}

__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                             const float* __restrict__ lse, const float* __restrict__ gloss,
                                                             float* __restrict__ dlogits, int B, int C, int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* x = logits + (long long)row * ld;
  float* d = dlogits + (long long)row * ld;
  const float l = lse[row], g = gloss[0] / (float)B;
  const long long y = labels[row];
  for (int c = lane; c < ld; c += 64) {
    float v = 0.f;
    if (c < C) v = (__expf(x[c] - l) - (c == y ? 1.f : 0.f)) * g;
    d[c] = v;   // padding columns receive a zero gradient
  }
}

extern "C" int maai_softmax_ce_fwd(const float* logits, const long long* labels, float* loss, float* lse, int B, int C, int ld,
                                   void* stream) {
  MAAI_CHECK_ARG(logits && labels && loss && lse && B > 0 && C > 0 && ld >= C, "softmax_ce_fwd: bad arguments");
  hipStream_t st = ST(stream);
  if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) {
    maai_set_error("softmax_ce_fwd: memset failed");
    return MAAI_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, labels, loss, lse, B, C, ld);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_softmax_ce_bwd(const float* logits, const long long* labels, const float* lse, const float* gloss, float* dlogits,
                                   int B, int C, int ld, void* stream) {
  MAAI_CHECK_ARG(logits && labels && lse && gloss && dlogits && B > 0 && C > 0 && ld >= C, "softmax_ce_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(stream), logits, labels, lse, gloss, dlogits, B, C, ld);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// LARC: squared norms of every tensor of a parameter group in one launch, trust-ratio rescaling in another.
// slots / block maps: the format of maai_adam_step_multi (m and v unused).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multi_sqnorm_kernel(const maai_adam_slot* __restrict__ slots, const int* __restrict__ block_slot,
                                                           const long long* __restrict__ block_first, double* __restrict__ norms) {
  __shared__ float red[2][4];
  const int sidx = block_slot[blockIdx.x];
  const maai_adam_slot s = slots[sidx];
  const long long i0 = block_first[blockIdx.x];
  float sp = 0.f, sg = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long i = i0 + u * 256 + threadIdx.x;
    if (i < s.n) {
      const float p = s.p[i], g = s.g[i];
      sp += p * p;
      sg += g * g;
    }
  }
  sp = wave_sum(sp);
  sg = wave_sum(sg);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sp;
    red[1][threadIdx.x >> 6] = sg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&norms[2 * sidx + 0], (double)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])));
    atomicAdd(&norms[2 * sidx + 1], (double)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
  }
}

__global__ __launch_bounds__(256) void larc_scale_kernel(const maai_adam_slot* __restrict__ slots, const int* __restrict__ block_slot,
                                                         const long long* __restrict__ block_first, const double* __restrict__ norms,
                                                         float trust, float lr, float wd, float eps, int clip) {
  const int sidx = block_slot[blockIdx.x];
  const maai_adam_slot s = slots[sidx];
  const float pn = (float)sqrt(norms[2 * sidx + 0]), gn = (float)sqrt(norms[2 * sidx + 1]);
  if (!(pn != 0.f && gn != 0.f)) return;   // apex/parallel/LARC.py: tensors with a zero norm are left alone
  float rate = trust * pn / (gn + pn * wd + eps);
  if (clip) rate = fminf(rate / lr, 1.f);
  float* g = const_cast<float*>(s.g);
  const long long i0 = block_first[blockIdx.x];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long i = i0 + u * 256 + threadIdx.x;
    if (i < s.n) g[i] = (g[i] + wd * s.p[i]) * rate;
  }
}

extern "C" int maai_multi_sqnorm(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                                 int nslots, double* norms, void* stream) {
  MAAI_CHECK_ARG(slots && block_slot && block_first && norms && nblocks > 0 && nslots > 0, "multi_sqnorm: bad arguments");
  hipStream_t st = ST(stream);
  if (hipMemsetAsync(norms, 0, sizeof(double) * 2 * nslots, st) != hipSuccess) {
    maai_set_error("multi_sqnorm: memset failed");
    return MAAI_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(multi_sqnorm_kernel, dim3(nblocks), dim3(256), 0, st, slots, block_slot, block_first, norms);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_larc_scale(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                               const double* norms, float trust_coefficient, float lr, float weight_decay, float eps, int clip,
                               void* stream) {
  MAAI_CHECK_ARG(slots && block_slot && block_first && norms && nblocks > 0 && lr > 0.f, "larc_scale: bad arguments");
  hipLaunchKernelGGL(larc_scale_kernel, dim3(nblocks), dim3(256), 0, ST(stream), slots, block_slot, block_first, norms,
                     trust_coefficient, lr, weight_decay, eps, clip);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
