// Layout, pooling, cast, optimiser and augmentation kernels (K1, K11, K12 of
// SURVEY §2.3).  All HBM-bound streaming kernels: coalesced, vectorised where
// the layout allows, grid-stride with a capped grid.
#include "common.h"
#include "maai_internal.h"

static inline unsigned cap_grid(long long n, int per_block = 256) {
  long long g = (n + per_block - 1) / per_block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}
#define ST(stream) reinterpret_cast<hipStream_t>(stream)

// ---------------------------------------------------------------------------
// SimCLR.py:24 view packing: K x [B,H,W,3] u8 -> [B,H,W,Cpad] T, channel k*3+c
// ---------------------------------------------------------------------------
struct ViewPtrs {
  const unsigned char* v[8];
};

template <typename T>
__global__ __launch_bounds__(256) void pack_views_kernel(ViewPtrs vp, int K, long long npix, int Cpad, T* __restrict__ out) {
  // one thread per (pixel, 16-byte chunk of output channels)
  constexpr int E = Vec16<T>::N;
  const int cpr = Cpad / E;
  const long long total = npix * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / cpr;
    const int c0 = (int)(i - pix * cpr) * E;
    float f[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = c0 + e;
      const int k = c / 3, cc = c - 3 * k;
      f[e] = (k < K) ? (float)vp.v[k][pix * 3 + cc] : 0.f;
    }
    Vec16<T> v;
    v.set(f);
    v.store(out + i * E);
  }
}

extern "C" int maai_pack_views_u8(const void* const* views_host, int K, int B, int H, int W, int Cpad, void* out, int dtype,
                                  void* stream) {
  MAAI_CHECK_ARG(views_host && out && K >= 1 && K <= 8 && B > 0 && H > 0 && W > 0, "pack_views_u8: bad arguments");
  MAAI_CHECK_ARG(Cpad >= 3 * K && Cpad % 8 == 0, "pack_views_u8: Cpad must be >= 3K and a multiple of 8");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "pack_views_u8: bad dtype");
  ViewPtrs vp;
  for (int k = 0; k < 8; ++k) vp.v[k] = k < K ? (const unsigned char*)views_host[k] : nullptr;
  const long long npix = (long long)B * H * W;
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL(pack_views_kernel<bf16_t>, dim3(cap_grid(npix * (Cpad / 8))), dim3(256), 0, ST(stream), vp, K, npix, Cpad, (bf16_t*)out);
  else
    hipLaunchKernelGGL(pack_views_kernel<float>, dim3(cap_grid(npix * (Cpad / 4))), dim3(256), 0, ST(stream), vp, K, npix, Cpad, (float*)out);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// stem operand for 3-channel input: out[n,h,w,kw*4+c] = x[n,h,w+kw-3,c]
// (kw < 7, c < 3; zero elsewhere).  32 channels = 4 chunks (bf16) / 8 (f32).
// ---------------------------------------------------------------------------
template <typename T, bool U8>
__global__ __launch_bounds__(256) void stem_unroll_kernel(const void* __restrict__ src, int B, int H, int W, T* __restrict__ out) {
  constexpr int E = Vec16<T>::N;
  constexpr int CPR = 32 / E;
  const long long npix = (long long)B * H * W;
  const long long total = npix * CPR;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / CPR;
    const int c0 = (int)(i - pix * CPR) * E;
    const int w = (int)(pix % W);
    const long long nh = pix / W;  // n*H + h
    float f[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int ch = c0 + e, kw = ch >> 2, c = ch & 3;
      const int ww = w + kw - 3;
      float v = 0.f;
      if (kw < 7 && c < 3 && ww >= 0 && ww < W) {
        if (U8) {
          v = (float)((const unsigned char*)src)[(nh * W + ww) * 3 + c];
        } else {
          const long long n = nh / H;
          const int h = (int)(nh - n * H);
          v = ((const float*)src)[((n * 3 + c) * H + h) * W + ww];
        }
      }
      f[e] = v;
    }
    Vec16<T> v;
    v.set(f);
    v.store(out + i * E);
  }
}

extern "C" int maai_stem_unroll_nchw_f32(const float* x, int B, int H, int W, void* out, int dtype, void* stream) {
  MAAI_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0, "stem_unroll_nchw_f32: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "stem_unroll: bad dtype");
  const long long npix = (long long)B * H * W;
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL((stem_unroll_kernel<bf16_t, false>), dim3(cap_grid(npix * 4)), dim3(256), 0, ST(stream), x, B, H, W, (bf16_t*)out);
  else
    hipLaunchKernelGGL((stem_unroll_kernel<float, false>), dim3(cap_grid(npix * 8)), dim3(256), 0, ST(stream), x, B, H, W, (float*)out);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_stem_unroll_u8(const void* view, int B, int H, int W, void* out, int dtype, void* stream) {
  MAAI_CHECK_ARG(view && out && B > 0 && H > 0 && W > 0, "stem_unroll_u8: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "stem_unroll: bad dtype");
  const long long npix = (long long)B * H * W;
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL((stem_unroll_kernel<bf16_t, true>), dim3(cap_grid(npix * 4)), dim3(256), 0, ST(stream), view, B, H, W, (bf16_t*)out);
  else
    hipLaunchKernelGGL((stem_unroll_kernel<float, true>), dim3(cap_grid(npix * 8)), dim3(256), 0, ST(stream), view, B, H, W, (float*)out);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// NCHW fp32 <-> NHWC T via a 32x32 LDS transpose tile (both sides coalesced)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int C, long long HW, int Cpad, T* __restrict__ out) {
  __shared__ float tile[32][33];
  const long long n = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k;
    const long long p = p0 + tx;
    tile[k][tx] = (c < C && p < HW) ? x[(n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long long p = p0 + k;
    const int c = c0 + tx;
    if (p < HW && c < Cpad) Store<T>::st(out + (n * HW + p) * Cpad + c, tile[tx][k]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, int C, long long HW, int Cpad, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const long long n = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const long long p = p0 + k;
    const int c = c0 + tx;
    tile[k][tx] = (p < HW && c < Cpad) ? Store<T>::ld(x + (n * HW + p) * Cpad + c) : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k;
    const long long p = p0 + tx;
    if (c < C && p < HW) out[(n * C + c) * HW + p] = tile[tx][k];
  }
}

extern "C" int maai_nchw_f32_to_nhwc(const float* x, int B, int C, int H, int W, int Cpad, void* out, int dtype, void* stream) {
  MAAI_CHECK_ARG(x && out && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_f32_to_nhwc: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "nchw_f32_to_nhwc: bad dtype");
  const long long HW = (long long)H * W;
  dim3 grid((unsigned)((HW + 31) / 32), (Cpad + 31) / 32, B);
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), x, C, HW, Cpad, (bf16_t*)out);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, ST(stream), x, C, HW, Cpad, (float*)out);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_nhwc_to_nchw_f32(const void* x, int B, int C, int H, int W, int Cpad, float* out, int dtype, void* stream) {
  MAAI_CHECK_ARG(x && out && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nhwc_to_nchw_f32: bad arguments");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "nhwc_to_nchw_f32: bad dtype");
  const long long HW = (long long)H * W;
  dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, B);
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)x, C, HW, Cpad, out);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)x, C, HW, Cpad, out);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_nchw_f32_from_nhwc_grad(const void* g, int B, int C, int H, int W, int Cpad, float* out, int dtype, void* stream) {
  return maai_nhwc_to_nchw_f32(g, B, C, H, W, Cpad, out, dtype, stream);
}

// ---------------------------------------------------------------------------
// adaptive average pool (H % PH == 0, W % PW == 0), NHWC, fp32 accumulate
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, int H, int W, int C, int PH, int PW,
                                                          T* __restrict__ out, long long total) {
  constexpr int E = Vec16<T>::N;
  const int cpr = C / E, wh = H / PH, ww = W / PW;
  const float inv = 1.f / (float)(wh * ww);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    long long t = i / cpr;
    const int pw = (int)(t % PW);
    t /= PW;
    const int ph = (int)(t % PH);
    const long long n = t / PH;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int a = 0; a < wh; ++a)
      for (int b = 0; b < ww; ++b) {
        Vec16<T> v;
        v.load(x + ((n * H + ph * wh + a) * W + pw * ww + b) * C + ch * E);
        float f[E];
        v.get(f);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += f[e];
      }
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] *= inv;
    Vec16<T> o;
    o.set(acc);
    o.store(out + i * E);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dout, int H, int W, int C, int PH, int PW,
                                                          T* __restrict__ dx, long long total) {
  constexpr int E = Vec16<T>::N;
  const int cpr = C / E, wh = H / PH, ww = W / PW;
  const float inv = 1.f / (float)(wh * ww);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    long long t = i / cpr;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    Vec16<T> v;
    v.load(dout + ((n * PH + h / wh) * PW + w / ww) * C + ch * E);
    float f[E];
    v.get(f);
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] *= inv;
    v.set(f);
    v.store(dx + i * E);
  }
}

extern "C" int maai_avgpool_fwd(const void* x, int B, int H, int W, int C, int PH, int PW, void* out, int dtype, void* stream) {
  MAAI_CHECK_ARG(x && out && B > 0 && PH > 0 && PW > 0 && H % PH == 0 && W % PW == 0, "avgpool_fwd: H, W must be multiples of the pooled size");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "avgpool_fwd: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "avgpool_fwd: C must be a multiple of the vector width");
  const long long total = (long long)B * PH * PW * (C / E);
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(cap_grid(total)), dim3(256), 0, ST(stream), (const bf16_t*)x, H, W, C, PH, PW, (bf16_t*)out, total);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(cap_grid(total)), dim3(256), 0, ST(stream), (const float*)x, H, W, C, PH, PW, (float*)out, total);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_avgpool_bwd(const void* dout, int B, int H, int W, int C, int PH, int PW, void* dx, int dtype, void* stream) {
  MAAI_CHECK_ARG(dout && dx && B > 0 && PH > 0 && PW > 0 && H % PH == 0 && W % PW == 0, "avgpool_bwd: H, W must be multiples of the pooled size");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "avgpool_bwd: bad dtype");
  const int E = dtype == MAAI_BF16 ? 8 : 4;
  MAAI_CHECK_ARG(C % E == 0, "avgpool_bwd: C must be a multiple of the vector width");
  const long long total = (long long)B * H * W * (C / E);
  if (dtype == MAAI_BF16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(cap_grid(total)), dim3(256), 0, ST(stream), (const bf16_t*)dout, H, W, C, PH, PW, (bf16_t*)dx, total);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(cap_grid(total)), dim3(256), 0, ST(stream), (const float*)dout, H, W, C, PH, PW, (float*)dx, total);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// casts
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_to_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long long n) {
  const long long nv = n / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(s)[2 * i], b = reinterpret_cast<const float4*>(s)[2 * i + 1];
    uint4 o;
    o.x = pack_bf16x2(a.x, a.y);
    o.y = pack_bf16x2(a.z, a.w);
    o.z = pack_bf16x2(b.x, b.y);
    o.w = pack_bf16x2(b.z, b.w);
    reinterpret_cast<uint4*>(d)[i] = o;
  }
  if (blockIdx.x == 0)
    for (long long i = nv * 8 + threadIdx.x; i < n; i += 256) d[i] = f32_to_bf16(s[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_to_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = bf16_to_f32(s[i]);
}

extern "C" int maai_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream) {
  MAAI_CHECK_ARG(src && dst && n > 0, "cast_from_f32: bad arguments");
  if (dtype == MAAI_F32) {
    if (hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, ST(stream)) != hipSuccess) return MAAI_ERR_LAUNCH;
    return MAAI_OK;
  }
  MAAI_CHECK_ARG(dtype == MAAI_BF16, "cast_from_f32: bad dtype");
  MAAI_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "cast_from_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(cap_grid(n / 8 + 1)), dim3(256), 0, ST(stream), src, (bf16_t*)dst, n);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
extern "C" int maai_cast_to_f32(const void* src, float* dst, long long n, int dtype, void* stream) {
  MAAI_CHECK_ARG(src && dst && n > 0, "cast_to_f32: bad arguments");
  if (dtype == MAAI_F32) {
    if (hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, ST(stream)) != hipSuccess) return MAAI_ERR_LAUNCH;
    return MAAI_OK;
  }
  MAAI_CHECK_ARG(dtype == MAAI_BF16, "cast_to_f32: bad dtype");
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(cap_grid(n)), dim3(256), 0, ST(stream), (const bf16_t*)src, dst, n);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// optimisers (torch.optim.Adam / SGD semantics, Model_Util.py:68-88)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float beta1,
                                                   float beta2, float omb1, float omb2, float eps, float inv_sqrt_bc2,
                                                   float gscale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = beta1 * m[i] + omb1 * gi;
    const float vi = beta2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

extern "C" int maai_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                              double eps, int step, float grad_scale, void* stream) {
  MAAI_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  // hyper-parameters arrive as doubles and are rounded to fp32 once, exactly where torch.optim.Adam rounds them
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(cap_grid(n)), dim3(256), 0, ST(stream), p, g, m, v, n, (float)(lr / bc1), (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// kernel-layout copies of the fp32 master weights (resnet.py:20-28 conv weights [Cout][Cin][KH][KW]), all layers in
// ONE launch after each optimiser step: forward form [Cout][a][b][Cin_pad] and data-gradient form [Cin][a][b][Cout]
// over a listed subset of taps (a, b index khs[] / kws[]; the stride-2 gradient's parity classes use subsets, the
// stride-1 gradient the reversed full list).  Built per layer with torch ops this was ~400 launches per step, each
// shorter than its issue time.  Block b converts 1024 output elements of form block_form[b] from block_first[b].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_forms_kernel(const maai_weight_form* __restrict__ forms, const int* __restrict__ block_form,
                                                           const long long* __restrict__ block_first) {
  const maai_weight_form f = forms[block_form[blockIdx.x]];
  const long long total = f.mode == 0 ? (long long)f.Cout * f.nkh * f.nkw * f.cin_pad : (long long)f.Cin * f.nkh * f.nkw * f.Cout;
  const int inner = f.mode == 0 ? f.cin_pad : f.Cout;
  const int kk = f.KH * f.KW;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = block_first[blockIdx.x] + u * 256 + threadIdx.x;
    if (i >= total) break;
    const int in = (int)(i % inner);
    long long r = i / inner;
    const int b = (int)(r % f.nkw);
    r /= f.nkw;
    const int a = (int)(r % f.nkh);
    const int outer = (int)(r / f.nkh);
    const int co = f.mode == 0 ? outer : in, ci = f.mode == 0 ? in : outer;
    float v = 0.f;
    if (ci < f.Cin) v = f.w[((long long)co * f.Cin + ci) * kk + f.khs[a] * f.KW + f.kws[b]];
    if (f.dtype == MAAI_BF16)
      reinterpret_cast<bf16_t*>(f.out)[i] = f32_to_bf16(v);
    else
      reinterpret_cast<float*>(f.out)[i] = v;
  }
}

extern "C" int maai_weight_forms(const maai_weight_form* forms, const int* block_form, const long long* block_first, int nblocks,
                                 void* stream) {
  MAAI_CHECK_ARG(forms && block_form && block_first && nblocks > 0, "weight_forms: bad arguments");
  hipLaunchKernelGGL(weight_forms_kernel, dim3(nblocks), dim3(256), 0, ST(stream), forms, block_form, block_first);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// all parameter tensors of a group in ONE launch: ~160 launches per step otherwise, each shorter than the time
// the host needs to issue it.  slots[] and the block map live in device memory; block b updates 2048 elements of
// tensor block_slot[b] starting at block_first[b].
__global__ __launch_bounds__(256) void adam_multi_kernel(const maai_adam_slot* __restrict__ slots, const int* __restrict__ block_slot,
                                                         const long long* __restrict__ block_first, float step_size, float beta1,
                                                         float beta2, float omb1, float omb2, float eps, float inv_sqrt_bc2,
                                                         float gscale) {
  const maai_adam_slot s = slots[block_slot[blockIdx.x]];
  const long long i0 = block_first[blockIdx.x];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long i = i0 + u * 256 + threadIdx.x;
    if (i < s.n) {
      const float gi = s.g[i] * gscale;
      const float mi = beta1 * s.m[i] + omb1 * gi;
      const float vi = beta2 * s.v[i] + omb2 * gi * gi;
      s.m[i] = mi;
      s.v[i] = vi;
      s.p[i] = s.p[i] - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
  }
}

extern "C" int maai_adam_step_multi(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                                    double lr, double beta1, double beta2, double eps, int step, float grad_scale, void* stream) {
  MAAI_CHECK_ARG(slots && block_slot && block_first && nblocks > 0 && step >= 1, "adam_step_multi: bad arguments");
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, ST(stream), slots, block_slot, block_first, (float)(lr / bc1),
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(1.0 / sqrt(bc2)),
                     grad_scale);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                                  long long n, float lr, float momentum, float wd, int first) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float d = g[i] + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? d : momentum * mom[i] + d;
      mom[i] = b;
      d = b;
    }
    p[i] -= lr * d;
  }
}

extern "C" int maai_sgd_step(float* p, const float* g, float* mom, long long n, float lr, float momentum, float weight_decay,
                             int first_step, void* stream) {
  MAAI_CHECK_ARG(p && g && n > 0 && (momentum == 0.f || mom), "sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(cap_grid(n)), dim3(256), 0, ST(stream), p, g, mom, n, lr, momentum, weight_decay, first_step);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// every tensor of a parameter group in one launch (slots: p, g, m = momentum buffer, v unused, n); arithmetic of sgd_kernel
__global__ __launch_bounds__(256) void sgd_multi_kernel(const maai_adam_slot* __restrict__ slots, const int* __restrict__ block_slot,
                                                        const long long* __restrict__ block_first, float lr, float momentum, float wd,
                                                        int first) {
  const maai_adam_slot s = slots[block_slot[blockIdx.x]];
  const long long i0 = block_first[blockIdx.x];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long long i = i0 + u * 256 + threadIdx.x;
    if (i < s.n) {
      float d = s.g[i] + wd * s.p[i];
      if (momentum != 0.f) {
        const float b = first ? d : momentum * s.m[i] + d;
        s.m[i] = b;
        d = b;
      }
      s.p[i] -= lr * d;
    }
  }
}

extern "C" int maai_sgd_step_multi(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                                   float lr, float momentum, float weight_decay, int first_step, void* stream) {
  MAAI_CHECK_ARG(slots && block_slot && block_first && nblocks > 0, "sgd_step_multi: bad arguments");
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(nblocks), dim3(256), 0, ST(stream), slots, block_slot, block_first, lr, momentum,
                     weight_decay, first_step);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// augmentation (replaces NVIDIA_DALI_Pipelines.py:444-480 for the north-star path)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ img, const float* __restrict__ params,
                                                      int H, int W, int OH, int OW, unsigned char* __restrict__ out, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const long long t = i / OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    const float* p = params + n * 16;
    const float x0 = p[0], y0 = p[1], cw = p[2], ch = p[3], flip = p[4], br = p[5], ct = p[6];
    const float xs = flip >= 0.5f ? (float)(OW - 1 - ox) : (float)ox;
    // written with explicit fp32 ops in the oracle's order (no fma contraction)
    const float fy = __fadd_rn(y0, __fmul_rn(__fadd_rn((float)oy, 0.5f), __fdiv_rn(ch, (float)OH)));
    const float fx = __fadd_rn(x0, __fmul_rn(__fadd_rn(xs, 0.5f), __fdiv_rn(cw, (float)OW)));
    int sy = (int)floorf(fy), sx = (int)floorf(fx);
    sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
    sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
    const unsigned char* s = img + ((n * H + sy) * W + sx) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = __fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn((float)s[c], 128.f), ct), 128.f), br);
    // hue rotation + saturation: one 3x3 colour matrix per sample (DALI ColorTwist's YIQ form), p[7..15] row major
    unsigned char* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* m = p + 7 + 3 * c;
      float r = __fadd_rn(__fadd_rn(__fmul_rn(m[0], v[0]), __fmul_rn(m[1], v[1])), __fmul_rn(m[2], v[2]));
      r = fminf(fmaxf(r, 0.f), 255.f);
      r = floorf(__fadd_rn(r, 0.5f));
      o[c] = (unsigned char)fminf(r, 255.f);
    }
  }
}

extern "C" int maai_augment_view_u8(const void* images, const float* params, int B, int H, int W, int OH, int OW, void* out, void* stream) {
  MAAI_CHECK_ARG(images && params && out && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "augment_view_u8: bad arguments");
  const long long total = (long long)B * OH * OW;
  hipLaunchKernelGGL(augment_kernel, dim3(cap_grid(total)), dim3(256), 0, ST(stream), (const unsigned char*)images, params, H, W, OH, OW,
                     (unsigned char*)out, total);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// counter-based hash RNG (splitmix64 finaliser) -> uniform [0,1)
__device__ __forceinline__ float u01(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void augment_params_kernel(float* __restrict__ params, int B, int H, int W, unsigned long long seed, int view,
                                      float min_area, float brightness, float contrast, float saturation, float hue) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= B) return;
  const unsigned long long base = ((unsigned long long)view << 40) + (unsigned long long)n * 16;
  // RandomResizedCrop(area in [min_area,1], aspect in [3/4,4/3]) (NVIDIA_DALI_Pipelines.py:416)
  const float area = (min_area + (1.f - min_area) * u01(seed, base + 0)) * (float)H * (float)W;
  const float logr = (u01(seed, base + 1) * 2.f - 1.f) * 0.28768207f;  // ln(4/3)
  const float ar = __expf(logr);
  float cw = sqrtf(area * ar), ch = sqrtf(area / ar);
  cw = fminf(cw, (float)W);
  ch = fminf(ch, (float)H);
  const float x0 = u01(seed, base + 2) * ((float)W - cw);
  const float y0 = u01(seed, base + 3) * ((float)H - ch);
  float* p = params + (long long)n * 16;
  p[0] = x0;
  p[1] = y0;
  p[2] = cw;
  p[3] = ch;
  p[4] = u01(seed, base + 4) < 0.5f ? 1.f : 0.f;  // CoinFlip(0.5) (NVIDIA_DALI_Pipelines.py:435)
  // Contrastive_Learning.py:622-630: b = (1 - B/2) + B*u, c likewise, hue = u*HUE degrees, s = (1 - S) + S*u
  p[5] = (1.f - brightness * 0.5f) + brightness * u01(seed, base + 5);
  p[6] = (1.f - contrast * 0.5f) + contrast * u01(seed, base + 6);
  const float sat = (1.f - saturation) + saturation * u01(seed, base + 7);
  const float hdeg = hue * u01(seed, base + 8);
  // colour matrix M = YIQ2RGB * R(hue) * diag(1, s, s) * RGB2YIQ (ColorTwist, NVIDIA_DALI_Pipelines.py:433,455-462)
  const float hr = hdeg * 0.017453292519943295f;
  const float cs = cosf(hr) * sat, sn = sinf(hr) * sat;
  const float A[3][3] = {{0.299f, 0.587f, 0.114f}, {0.596f, -0.274f, -0.321f}, {0.211f, -0.523f, 0.311f}};   // RGB -> YIQ
  const float Bm[3][3] = {{1.f, 0.956f, 0.621f}, {1.f, -0.272f, -0.647f}, {1.f, -1.107f, 1.705f}};           // YIQ -> RGB
  float T[3][3];  // R(hue) * diag(1, s, s) * A
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    T[0][j] = A[0][j];
    T[1][j] = cs * A[1][j] - sn * A[2][j];
    T[2][j] = sn * A[1][j] + cs * A[2][j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) p[7 + 3 * i + j] = Bm[i][0] * T[0][j] + Bm[i][1] * T[1][j] + Bm[i][2] * T[2][j];
}

extern "C" int maai_augment_params(float* params, int B, int H, int W, unsigned long long seed, int view, float min_area,
                                   float brightness, float contrast, float saturation, float hue, void* stream) {
  MAAI_CHECK_ARG(params && B > 0 && H > 0 && W > 0 && min_area > 0.f && min_area <= 1.f, "augment_params: bad arguments");
  hipLaunchKernelGGL(augment_params_kernel, dim3((B + 255) / 256), dim3(256), 0, ST(stream), params, B, H, W, seed, view, min_area,
                     brightness, contrast, saturation, hue);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
