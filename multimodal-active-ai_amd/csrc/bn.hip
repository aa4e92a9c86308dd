// BatchNorm (training / eval) + ReLU + residual, forward and backward, NHWC.
// K5/K6 of SURVEY §2.3: nn.BatchNorm2d / nn.SyncBatchNorm as norm_layer in
// resnet.py:54-55,72-75,106-133.  All of these are HBM-bound streaming passes:
// 16 bytes per lane per access, channel parameters from L1/L2, fp32 maths.
// Statistics come from the conv epilogue (conv_fwd.hip) as fp32 partial slabs
// and are summed in fp64 here, so cross-rank SyncBN only has to all-reduce the
// 2C doubles between maai_reduce_partials and maai_bn_finalize.
#include "common.h"
#include "maai_internal.h"

// ---------------------------------------------------------------------------
// partial[rows][C2] fp32 -> sums[C2] fp64 (atomic fp64 adds of per-block sums).
// The slabs are large at the full-resolution stages (one row per 128/256-pixel conv tile: 100k rows x 2C floats),
// so this is a streaming pass like the others: 16 bytes per lane, L lanes per row (L = the largest power of two
// <= 64 dividing C2/4), 256/L rows per pass, ~1k workgroups, four independent fp64 accumulators per lane.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, long long rows, int C2,
                                                              double* __restrict__ sums, long long rows_per_block, int L) {
  __shared__ double red[256][4];
  const int lr = threadIdx.x % L, slot = threadIdx.x / L, nslot = 256 / L;
  const int col = (blockIdx.x * L + lr) * 4;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const float* p = partial + col;
  long long r = r0 + slot;
  for (; r + 7 * nslot < r1; r += 8 * nslot) {  // eight rows (128 bytes per lane) in flight: the slabs of the full-resolution
    float4 v[8];                                //  stages are hundreds of MB, streamed once (non-temporal)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint4 u = ld16_nt(p + (r + k * nslot) * C2);
      v[k] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    }
    s0 += (((double)v[0].x + (double)v[1].x) + ((double)v[2].x + (double)v[3].x)) + (((double)v[4].x + (double)v[5].x) + ((double)v[6].x + (double)v[7].x));
    s1 += (((double)v[0].y + (double)v[1].y) + ((double)v[2].y + (double)v[3].y)) + (((double)v[4].y + (double)v[5].y) + ((double)v[6].y + (double)v[7].y));
    s2 += (((double)v[0].z + (double)v[1].z) + ((double)v[2].z + (double)v[3].z)) + (((double)v[4].z + (double)v[5].z) + ((double)v[6].z + (double)v[7].z));
    s3 += (((double)v[0].w + (double)v[1].w) + ((double)v[2].w + (double)v[3].w)) + (((double)v[4].w + (double)v[5].w) + ((double)v[6].w + (double)v[7].w));
  }
  for (; r + 3 * nslot < r1; r += 4 * nslot) {  // four rows in flight
    const float4 a = *reinterpret_cast<const float4*>(p + r * C2);
    const float4 b = *reinterpret_cast<const float4*>(p + (r + nslot) * C2);
    const float4 c = *reinterpret_cast<const float4*>(p + (r + 2 * nslot) * C2);
    const float4 d = *reinterpret_cast<const float4*>(p + (r + 3 * nslot) * C2);
    s0 += ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x);
    s1 += ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
    s2 += ((double)a.z + (double)b.z) + ((double)c.z + (double)d.z);
    s3 += ((double)a.w + (double)b.w) + ((double)c.w + (double)d.w);
  }
  for (; r < r1; r += nslot) {
    const float4 a = *reinterpret_cast<const float4*>(p + r * C2);
    s0 += (double)a.x;
    s1 += (double)a.y;
    s2 += (double)a.z;
    s3 += (double)a.w;
  }
  red[threadIdx.x][0] = s0;
  red[threadIdx.x][1] = s1;
  red[threadIdx.x][2] = s2;
  red[threadIdx.x][3] = s3;
  __syncthreads();
  // thread t < 4L: column (t/4 of this block's L groups, component t%4), summed over the row slots
  if (threadIdx.x < 4 * L) {
    const int g = threadIdx.x >> 2, e = threadIdx.x & 3;
    double t = 0.0;
    for (int k = 0; k < nslot; ++k) t += red[k * L + g][e];
    atomicAdd(&sums[(blockIdx.x * L + g) * 4 + e], t);
  }
}

// scalar fallback for column counts that are not a multiple of 4
__global__ __launch_bounds__(256) void reduce_partials_scalar_kernel(const float* __restrict__ partial, long long rows, int C2,
                                                                     double* __restrict__ sums, long long rows_per_block) {
  __shared__ double red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  double s = 0.0;
  if (col < C2)
    for (long long r = r0 + rg; r < r1; r += 4) s += (double)partial[r * C2 + col];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && col < C2) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&sums[col], s);
  }
}

extern "C" int maai_reduce_partials(const float* partial, long long rows, int C2, double* sums, void* stream) {
  MAAI_CHECK_ARG(partial && sums && rows > 0 && C2 > 0, "reduce_partials: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(sums, 0, sizeof(double) * C2, st) != hipSuccess) {
    maai_set_error("reduce_partials: memset failed");
    return MAAI_ERR_LAUNCH;
  }
  if (C2 % 4 == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0) {
    const int c4 = C2 / 4;
    int L = c4 & -c4;  // largest power of two dividing c4
    if (L > 64) L = 64;
    const int gx = c4 / L, nslot = 256 / L;
    long long slices = (rows + 8LL * nslot - 1) / (8LL * nslot);  // >= 8 rows per row slot
    const long long cap = 2048 / gx > 0 ? 2048 / gx : 1;
    if (slices > cap) slices = cap;
    if (slices < 1) slices = 1;
    const long long rpb = (rows + slices - 1) / slices;
    dim3 grid(gx, (unsigned)((rows + rpb - 1) / rpb));
    MAAI_NOTE_KERNEL(reduce_partials_kernel);
    hipLaunchKernelGGL(reduce_partials_kernel, grid, dim3(256), 0, st, partial, rows, C2, sums, rpb, L);
  } else {
    long long slices = (rows + 255) / 256;
    if (slices > 64) slices = 64;
    const long long rpb = (rows + slices - 1) / slices;
    dim3 grid((C2 + 63) / 64, (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(reduce_partials_scalar_kernel, grid, dim3(256), 0, st, partial, rows, C2, sums, rpb);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// finalize: torch.nn.BatchNorm2d training semantics
// ---------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* mean_o, float* invstd_o, float* scale_o,
                                   float* shift_o, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = (float)((double)g * invstd);
  if (mean_o) mean_o[c] = (float)mean;
  if (invstd_o) invstd_o[c] = (float)invstd;
  scale_o[c] = sc;
  shift_o[c] = (float)((double)b - mean * (double)g * invstd);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

extern "C" int maai_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                float* invstd, float* scale, float* shift, int C, void* stream) {
  MAAI_CHECK_ARG(sums && scale && shift && C > 0 && count > 0, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums,
                     count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// Deferred running-statistic updates (engine: the two forwards of a SimCLR step on two streams).  A forward that must not
// touch the running buffers while another forward is in flight takes its (float)mean / (float)unbiased variance into
// scratch (maai_bn_finalize with momentum 1 on a zeroed scratch buffer writes exactly those values); after the streams have
// joined, ONE launch applies the updates of every layer in program order — stat_a (the earlier forward's) then stat_b —
// with the arithmetic of maai_bn_finalize, (1 - momentum)*running + momentum*stat: the buffers end up bit-identical to two
// forwards run one after the other.
struct BnUpdateSlot {
  float* running;
  const float* stat_a;
  const float* stat_b;   // nullable
  long long n;
  float momentum;
  int pad;
};
__global__ __launch_bounds__(256) void bn_running_update_kernel(const BnUpdateSlot* __restrict__ slots) {
  const BnUpdateSlot s = slots[blockIdx.x];
  for (long long i = threadIdx.x; i < s.n; i += 256) {
    float r = s.running[i];
    r = (1.f - s.momentum) * r + s.momentum * s.stat_a[i];
    if (s.stat_b) r = (1.f - s.momentum) * r + s.momentum * s.stat_b[i];
    s.running[i] = r;
  }
}
static_assert(sizeof(BnUpdateSlot) == sizeof(maai_bn_update_slot), "slot layout");
extern "C" int maai_bn_running_update_multi(const maai_bn_update_slot* slots, int nslots, void* stream) {
  MAAI_CHECK_ARG(slots && nslots > 0, "bn_running_update_multi: bad arguments");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((unsigned)nslots), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const BnUpdateSlot*>(slots));
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// SyncBatchNorm exchange in fp32: every rank sends  mean[C] | M2[C] | count  (M2 = sum (x - mean)^2 over ITS samples; the
// count travels as the bit pattern of an int32, gathers do no arithmetic) — 2C+1 words, all well scaled, unlike the raw
// sums of squares — and every rank merges the gathered rows with Chan's parallel-variance formula in fp64:
//   N = sum n_r,  mean = sum n_r mean_r / N,  M2 = sum (M2_r + n_r (mean_r - mean)^2),  var = M2 / N.
// (nn.SyncBatchNorm gathers mean | invstd | count per layer the same way; Contrastive_Learning.py:240-252 selects it.)
// ---------------------------------------------------------------------------
__global__ void bn_pack_stats_kernel(const double* __restrict__ sums, double count, float* __restrict__ packed, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  if (c == C) {
    packed[2 * C] = __int_as_float((int)count);
    return;
  }
  const double mean = sums[c] / count;
  double m2 = sums[C + c] - count * mean * mean;
  if (m2 < 0.0) m2 = 0.0;
  packed[c] = (float)mean;
  packed[C + c] = (float)m2;
}

extern "C" int maai_bn_pack_stats(const double* sums, double count, float* packed, int C, void* stream) {
  MAAI_CHECK_ARG(sums && packed && C > 0 && count > 0 && count < 2147483648.0, "bn_pack_stats: bad arguments");
  hipLaunchKernelGGL(bn_pack_stats_kernel, dim3((C + 256) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums, count,
                     packed, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

__global__ void bn_finalize_gathered_kernel(const float* __restrict__ g, int world, long long row_stride,
                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                            float* running_var, float momentum, float eps, float* mean_o, float* invstd_o,
                                            float* scale_o, float* shift_o, double* count_o, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double N = 0.0, S = 0.0;
  for (int r = 0; r < world; ++r) {
    const double n = (double)__float_as_int(g[r * row_stride + 2 * C]);
    N += n;
    S += n * (double)g[r * row_stride + c];
  }
  if (count_o && c == 0) *count_o = N;   // the merged sample count: what the backward's 1/N uses (the ranks' batches may differ)
  const double mean = S / N;
  double M2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double n = (double)__float_as_int(g[r * row_stride + 2 * C]);
    const double d = (double)g[r * row_stride + c] - mean;
    M2 += (double)g[r * row_stride + C + c] + n * d * d;
  }
  double var = M2 / N;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float gm = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  if (mean_o) mean_o[c] = (float)mean;
  if (invstd_o) invstd_o[c] = (float)invstd;
  scale_o[c] = (float)((double)gm * invstd);
  shift_o[c] = (float)((double)b - mean * (double)gm * invstd);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unb = N > 1.0 ? var * (N / (N - 1.0)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

extern "C" int maai_bn_finalize_gathered(const float* gathered, int world, long long row_stride, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                         float* mean, float* invstd, float* scale, float* shift, double* count_out, int C,
                                         void* stream) {
  MAAI_CHECK_ARG(gathered && scale && shift && C > 0 && world > 0 && row_stride >= 2LL * C + 1, "bn_finalize_gathered: bad arguments");
  hipLaunchKernelGGL(bn_finalize_gathered_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gathered, world, row_stride, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                     count_out, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

extern "C" int maai_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift, int C, void* stream) {
  MAAI_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_eval_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gamma, beta, running_mean, running_var, eps, scale, shift, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// forward apply: out = act(y*scale + shift (+ residual))
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res,
                                                         T* __restrict__ out, unsigned char* __restrict__ bits,
                                                         long long nchunks, int cpr, int relu) {
  constexpr int E = Vec16<T>::N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr) * E;
    Vec16<T> v;
    v.load_nt(y + i * E);
    float f[E];
    v.get(f);
    if (scale) {
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] *= scale[c + e];
    }
    if (shift) {
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] += shift[c + e];
    }
    if (res) {
      Vec16<T> r;
      r.load_nt(res + i * E);
      float g[E];
      r.get(g);
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] += g[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    v.set(f);
    v.store_nt(out + i * E);
    if (E == 8 && bits) {  // 1-bit ReLU mask of the STORED value: byte i = elements 8i..8i+7, bit e = (out > 0)
      v.get(f);
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
      bits[i] = (unsigned char)b;
    }
  }
}

// The first block of a stage adds two normalised branches (resnet.py:126-133 with a downsample: out =
// relu(bn3(y3) + bn_d(y_d))).  Doing it in one pass — out = act((y*s + t) + r(y2*s2 + t2)) — saves writing the
// normalised shortcut and reading it back (8 of 20 tensor passes).  r() rounds the shortcut to the storage type
// first, so the result is bit-identical to the two-pass sequence it replaces.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd2_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ y2,
                                                          const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                          T* __restrict__ out, unsigned char* __restrict__ bits,
                                                          long long nchunks, int cpr, int relu) {
  constexpr int E = Vec16<T>::N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr) * E;
    Vec16<T> v, w;
    v.load_nt(y + i * E);
    w.load_nt(y2 + i * E);
    float f[E], g[E];
    v.get(f);
    w.get(g);
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = g[e] * scale2[c + e] + shift2[c + e];
    w.set(g);
    w.get(g);  // the shortcut as the two-pass path would have stored it
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] = (f[e] * scale[c + e] + shift[c + e]) + g[e];
    if (relu) {
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    v.set(f);
    v.store_nt(out + i * E);
    if (E == 8 && bits) {
      v.get(f);
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
      bits[i] = (unsigned char)b;
    }
  }
}

static inline unsigned stream_grid(long long nchunks) {
  // Streamed-once tensors use non-temporal accesses and a 32768-block grid: measured on MI355X
  // (scripts/probes/stream_probe.hip, bf16 BatchNorm + residual + ReLU on 6.6 GB tensors, one process) 5.79 TB/s as
  // plain accesses from 8192 blocks -> 5.96 non-temporal -> 6.25 from 32768 blocks; hoisting the per-channel
  // coefficients out of the loop LOST 3 % and the 1-bit mask's byte stores cost 2.5 %.
  long long g = (nchunks + 255) / 256;
  if (g > 32768) g = 32768;
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" int maai_bn_act_fwd(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                               long long M, int C, int relu, int dtype, void* stream) {
  return maai_bn_act_fwd_mask(y, scale, shift, residual, out, nullptr, M, C, relu, dtype, stream);
}

extern "C" int maai_bn_act_fwd_mask(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                                    unsigned char* mask_bits, long long M, int C, int relu, int dtype, void* stream) {
  MAAI_CHECK_ARG(y && out && M > 0 && C > 0, "bn_act_fwd: bad arguments");
  MAAI_CHECK_ARG(!mask_bits || dtype == MAAI_BF16, "bn_act_fwd: the 1-bit mask is produced for bf16 tensors only");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "bn_act_fwd: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "bn_act_fwd: C must be a multiple of 8 (bf16) / 4 (f32)");
  const long long nchunks = M * (C / E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MAAI_BF16) {
    MAAI_NOTE_KERNEL(bn_act_fwd_kernel<bf16_t>);
    hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const bf16_t*)y, scale,
                       shift, (const bf16_t*)residual, (bf16_t*)out, mask_bits, nchunks, C / E, relu);
  }
  else {
    MAAI_NOTE_KERNEL(bn_act_fwd_kernel<float>);
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const float*)y, scale,
                       shift, (const float*)residual, (float*)out, nullptr, nchunks, C / E, relu);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// backward pass 1: per-channel sums of dz and dz*(y-mean)
// ---------------------------------------------------------------------------
static inline void bwd_geometry(long long M, int C, int dtype, int* cs, int* ny, long long* rpb, long long* rows) {
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  const int cpr = C / E;
  *cs = cpr < 256 ? cpr : 256;  // chunks (threads along C) per block slab
  *ny = (cpr + *cs - 1) / *cs;
  const int rpp = 256 / *cs;  // rows per pass
  long long r = (M + 2047) / 2048;
  if (r < 8 * rpp) r = 8 * rpp;
  r = (r + rpp - 1) / rpp * rpp;
  *rpb = r;
  *rows = (M + r - 1) / r;
}

extern "C" long long maai_bn_bwd_rows(long long M, int C, int dtype) {
  int cs, ny;
  long long rpb, rows;
  bwd_geometry(M, C, dtype, &cs, &ny, &rpb, &rows);
  return rows;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            float* __restrict__ partial, long long M, int C, int cs,
                                                            long long rpb, int relu) {
  constexpr int E = Vec16<T>::N;
  __shared__ float red[256 * 2 * E];
  const int tc = threadIdx.x % cs, tr = threadIdx.x / cs, rpp = 256 / cs;
  const int chunk = blockIdx.y * cs + tc;
  const int c = chunk * E;
  const bool cok = c < C;
  float s1[E], s2[E], mu[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    s1[e] = 0.f;
    s2[e] = 0.f;
    mu[e] = (mean && cok) ? mean[c + e] : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * rpb;
  long long r1 = r0 + rpb;
  if (r1 > M) r1 = M;
  if (cok) {
    for (long long r = r0 + tr; r < r1; r += rpp) {
      const long long off = r * C + c;
      Vec16<T> vd;
      vd.load_nt(dout + off);
      float d[E];
      vd.get(d);
      if (relu) {
        Vec16<T> vo;
        vo.load_nt(out + off);
        float o[E];
        vo.get(o);
#pragma unroll
        for (int e = 0; e < E; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) s1[e] += d[e];
      if (y) {
        Vec16<T> vy;
        vy.load_nt(y + off);
        float yy[E];
        vy.get(yy);
#pragma unroll
        for (int e = 0; e < E; ++e) s2[e] += d[e] * (yy[e] - mu[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    red[(threadIdx.x * 2 + 0) * E + e] = s1[e];
    red[(threadIdx.x * 2 + 1) * E + e] = s2[e];
  }
  __syncthreads();
  // thread t < cs*2*E sums over the rpp row-threads
  for (int o = threadIdx.x; o < cs * 2 * E; o += 256) {
    const int tcc = o / (2 * E), rem = o - tcc * 2 * E, which = rem / E, e = rem - which * E;
    float s = 0.f;
    for (int k = 0; k < rpp; ++k) s += red[((k * cs + tcc) * 2 + which) * E + e];
    const int cc = (blockIdx.y * cs + tcc) * E + e;
    if (cc < C) partial[((long long)blockIdx.x * 2 + which) * C + cc] = s;
  }
}

extern "C" int maai_bn_act_bwd_reduce(const void* dout, const void* out, const void* y, const float* mean,
                                      float* partial, long long M, int C, int relu, int dtype, void* stream) {
  MAAI_CHECK_ARG(dout && partial && M > 0 && C > 0, "bn_act_bwd_reduce: bad arguments");
  MAAI_CHECK_ARG(!relu || out, "bn_act_bwd_reduce: relu needs the forward output");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "bn_act_bwd_reduce: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "bn_act_bwd_reduce: C must be a multiple of 8 (bf16) / 4 (f32)");
  int cs, ny;
  long long rpb, rows;
  bwd_geometry(M, C, dtype, &cs, &ny, &rpb, &rows);
  MAAI_CHECK_ARG(256 % cs == 0, "bn_act_bwd_reduce: C/vector must divide 256 or be a multiple of it");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)rows, ny);
  if (dtype == MAAI_BF16) {
    MAAI_NOTE_KERNEL(bn_bwd_reduce_kernel<bf16_t>);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                       (const bf16_t*)y, mean, partial, M, C, cs, rpb, relu);
  }
  else {
    MAAI_NOTE_KERNEL(bn_bwd_reduce_kernel<float>);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)out,
                       (const float*)y, mean, partial, M, C, cs, rpb, relu);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

template <typename TS>   // TS: double (local / fp64-reduced sums) or float (the fp32 cross-rank exchange)
__global__ void bn_bwd_coeffs_kernel(const TS* __restrict__ sums, double count, const double* __restrict__ count_dev,
                                     const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                     float* k1, float* k2, float* k3, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = *count_dev;   // (SyncBatchNorm: the merged count of maai_bn_finalize_gathered, never read back by the host)
  const double S1 = sums[c], S2 = sums[C + c];
  const double is = invstd[c], g = gamma ? gamma[c] : 1.0, mu = mean[c];
  if (dbeta) dbeta[c] = (float)S1;
  if (dgamma) dgamma[c] = (float)(is * S2);
  const double a = g * is;
  const double c3 = a * is * is * S2 / count;
  k1[c] = (float)a;
  k3[c] = (float)c3;
  k2[c] = (float)(a * S1 / count - c3 * mu);
}

extern "C" int maai_bn_bwd_coeffs(const double* sums, double count, const float* gamma, const float* mean,
                                  const float* invstd, float* dgamma, float* dbeta, float* k1, float* k2, float* k3, int C,
                                  const double* count_dev, void* stream) {
  MAAI_CHECK_ARG(sums && mean && invstd && k1 && k2 && k3 && C > 0 && (count > 0 || count_dev), "bn_bwd_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel<double>, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums,
                     count, count_dev, gamma, mean, invstd, dgamma, dbeta, k1, k2, k3, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_bn_bwd_coeffs_f32(const float* sums, double count, const float* gamma, const float* mean,
                                      const float* invstd, float* dgamma, float* dbeta, float* k1, float* k2, float* k3, int C,
                                      const double* count_dev, void* stream) {
  MAAI_CHECK_ARG(sums && mean && invstd && k1 && k2 && k3 && C > 0 && (count > 0 || count_dev), "bn_bwd_coeffs_f32: bad arguments");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel<float>, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums,
                     count, count_dev, gamma, mean, invstd, dgamma, dbeta, k1, k2, k3, C);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// backward pass 2: dy = k1*dz - k2 - k3*y  (and dz for the residual branch)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                           const T* __restrict__ y, const float* __restrict__ k1,
                                                           const float* __restrict__ k2, const float* __restrict__ k3,
                                                           T* __restrict__ dy, T* __restrict__ dz_out, long long nchunks,
                                                           int cpr, int relu) {
  constexpr int E = Vec16<T>::N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr) * E;
    Vec16<T> vd;
    vd.load_nt(dout + i * E);
    float d[E];
    vd.get(d);
    if (relu) {
      Vec16<T> vo;
      vo.load_nt(out + i * E);
      float o[E];
      vo.get(o);
#pragma unroll
      for (int e = 0; e < E; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
    }
    if (dz_out) {
      Vec16<T> vz;
      vz.set(d);
      vz.store_nt(dz_out + i * E);
    }
    if (dy) {
      if (k1) {
        Vec16<T> vy;
        vy.load_nt(y + i * E);
        float yy[E];
        vy.get(yy);
#pragma unroll
        for (int e = 0; e < E; ++e) d[e] = k1[c + e] * d[e] - k2[c + e] - k3[c + e] * yy[e];
      }
      Vec16<T> vr;
      vr.set(d);
      vr.store_nt(dy + i * E);
    }
  }
}

extern "C" int maai_bn_act_bwd_apply(const void* dout, const void* out, const void* y, const float* k1, const float* k2,
                                     const float* k3, void* dy, void* dz_out, long long M, int C, int relu, int dtype,
                                     void* stream) {
  MAAI_CHECK_ARG(dout && (dy || dz_out) && M > 0 && C > 0, "bn_act_bwd_apply: bad arguments");
  MAAI_CHECK_ARG(!relu || out, "bn_act_bwd_apply: relu needs the forward output");
  MAAI_CHECK_ARG(!k1 || (k2 && k3 && y), "bn_act_bwd_apply: k1 needs k2, k3 and y");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "bn_act_bwd_apply: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "bn_act_bwd_apply: C must be a multiple of 8 (bf16) / 4 (f32)");
  const long long nchunks = M * (C / E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MAAI_BF16) {
    MAAI_NOTE_KERNEL(bn_bwd_apply_kernel<bf16_t>);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const bf16_t*)dout,
                       (const bf16_t*)out, (const bf16_t*)y, k1, k2, k3, (bf16_t*)dy, (bf16_t*)dz_out, nchunks, C / E, relu);
  }
  else {
    MAAI_NOTE_KERNEL(bn_bwd_apply_kernel<float>);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const float*)dout,
                       (const float*)out, (const float*)y, k1, k2, k3, (float*)dy, (float*)dz_out, nchunks, C / E, relu);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_bn_act_fwd2(const void* y, const float* scale, const float* shift, const void* y2, const float* scale2,
                                const float* shift2, void* out, unsigned char* mask_bits, long long M, int C, int relu,
                                int dtype, void* stream) {
  MAAI_CHECK_ARG(y && y2 && scale && shift && scale2 && shift2 && out && M > 0 && C > 0, "bn_act_fwd2: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "bn_act_fwd2: bad dtype");
  MAAI_CHECK_ARG(!mask_bits || dtype == MAAI_BF16, "bn_act_fwd2: the 1-bit mask is produced for bf16 tensors only");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "bn_act_fwd2: C must be a multiple of 8 (bf16) / 4 (f32)");
  const long long nchunks = M * (C / E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MAAI_BF16) {
    MAAI_NOTE_KERNEL(bn_act_fwd2_kernel<bf16_t>);
    hipLaunchKernelGGL(bn_act_fwd2_kernel<bf16_t>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const bf16_t*)y, scale,
                       shift, (const bf16_t*)y2, scale2, shift2, (bf16_t*)out, mask_bits, nchunks, C / E, relu);
  }
  else {
    MAAI_NOTE_KERNEL(bn_act_fwd2_kernel<float>);
    hipLaunchKernelGGL(bn_act_fwd2_kernel<float>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const float*)y, scale,
                       shift, (const float*)y2, scale2, shift2, (float*)out, nullptr, nchunks, C / E, relu);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// backward of the same pair: both branches receive the same gradient dz, read once
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                            const float* __restrict__ k1, const float* __restrict__ k2,
                                                            const float* __restrict__ k3, const T* __restrict__ y2,
                                                            const float* __restrict__ k1b, const float* __restrict__ k2b,
                                                            const float* __restrict__ k3b, T* __restrict__ dy,
                                                            T* __restrict__ dy2, long long nchunks, int cpr) {
  constexpr int E = Vec16<T>::N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr) * E;
    Vec16<T> vd, va, vb;
    vd.load_nt(dz + i * E);
    va.load_nt(y + i * E);
    vb.load_nt(y2 + i * E);
    float d[E], a[E], b[E];
    vd.get(d);
    va.get(a);
    vb.get(b);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      a[e] = k1[c + e] * d[e] - k2[c + e] - k3[c + e] * a[e];
      b[e] = k1b[c + e] * d[e] - k2b[c + e] - k3b[c + e] * b[e];
    }
    va.set(a);
    vb.set(b);
    va.store_nt(dy + i * E);
    vb.store_nt(dy2 + i * E);
  }
}

extern "C" int maai_bn_act_bwd_apply2(const void* dz, const void* y, const float* k1, const float* k2, const float* k3,
                                      const void* y2, const float* k1b, const float* k2b, const float* k3b, void* dy,
                                      void* dy2, long long M, int C, int dtype, void* stream) {
  MAAI_CHECK_ARG(dz && y && y2 && k1 && k2 && k3 && k1b && k2b && k3b && dy && dy2 && M > 0 && C > 0,
                 "bn_act_bwd_apply2: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "bn_act_bwd_apply2: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "bn_act_bwd_apply2: C must be a multiple of 8 (bf16) / 4 (f32)");
  const long long nchunks = M * (C / E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MAAI_BF16) {
    MAAI_NOTE_KERNEL(bn_bwd_apply2_kernel<bf16_t>);
    hipLaunchKernelGGL(bn_bwd_apply2_kernel<bf16_t>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const bf16_t*)dz,
                       (const bf16_t*)y, k1, k2, k3, (const bf16_t*)y2, k1b, k2b, k3b, (bf16_t*)dy, (bf16_t*)dy2, nchunks, C / E);
  }
  else {
    MAAI_NOTE_KERNEL(bn_bwd_apply2_kernel<float>);
    hipLaunchKernelGGL(bn_bwd_apply2_kernel<float>, dim3(stream_grid(nchunks)), dim3(256), 0, st, (const float*)dz,
                       (const float*)y, k1, k2, k3, (const float*)y2, k1b, k2b, k3b, (float*)dy, (float*)dy2, nchunks, C / E);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
