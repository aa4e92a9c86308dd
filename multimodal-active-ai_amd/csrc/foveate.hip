// Foveated retinal processor (replaces the DALI graph of NVIDIA_DALI_Pipelines.py:444-480 on ROCm, where DALI
// does not exist): RandomResizedCrop(640x640) -> Rotate(angle, zero fill, centre-cropped back to 640) ->
// GridMask -> + Gaussian-like noise -> horizontal Flip -> ColorTwist -> 4 concentric crops (400/240/100/30 px
// at the fixation point) -> each resized to 30x30.  The reference launches ~12 DALI kernels over 640x640
// intermediates per fixation; here ONE kernel computes every 30x30x3x4 output pixel by mapping it backwards
// through the whole chain (the intermediates never exist), one thread per output pixel.
//
// All parameters arrive per sample as 32 floats (host-computed sines / colour matrix, so the device does no
// transcendental maths and the numpy restatement in oracle/ can match it to 1 LSB):
//   0 src_h 1 src_w | 2 x0 3 y0 4 cw 5 ch (crop window in source pixels) | 6 cos 7 sin (rotation) | 8 flip |
//   9 gm_ratio 10 gm_tile 11 gm_shift_x 12 gm_shift_y 13 gm_cos 14 gm_sin | 15 noise_mean 16 noise_std
//   17 noise_seed | 18..26 colour matrix M (row major) 27 gain 28 offset (v' = gain*(M v) + offset) |
//   29 pos_x 30 pos_y (fixation, 0..1)
// DALI itself is not installable here, so parity with DALI's exact filters is UNPINNED; the pinned
// contract is oracle/simclr_oracle.py::foveate_views.
#include "common.h"
#include "maai_internal.h"

#define FOV_CANVAS 640.0f

__device__ __forceinline__ unsigned fov_hash(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u + 0x85EBCA6Bu;
  h ^= b + 0x9E3779B9u + (h << 6) + (h >> 2);
  h ^= c + 0x9E3779B9u + (h << 6) + (h >> 2);
  h ^= d + 0x9E3779B9u + (h << 6) + (h >> 2);
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}

// value of the 640x640 canvas (after rotate, grid-mask, noise, flip, colour twist) at continuous (cx, cy)
__device__ __forceinline__ void fov_canvas(const unsigned char* __restrict__ img, int H, int W, const float* __restrict__ p,
                                           float cx, float cy, unsigned pix_id, float* rgb) {
  // flip acts after grid-mask and noise: evaluate those at the pre-flip coordinate
  const float fx = p[8] >= 0.5f ? __fsub_rn(FOV_CANVAS - 1.0f, cx) : cx;
  const float fy = cy;
  float v[3] = {0.f, 0.f, 0.f};
  // grid mask: zero inside the black square of each (rotated, shifted) tile
  bool masked = false;
  if (p[9] > 0.f) {
    const float gx = __fadd_rn(__fsub_rn(__fmul_rn(p[13], fx), __fmul_rn(p[14], fy)), p[11]);
    const float gy = __fadd_rn(__fadd_rn(__fmul_rn(p[14], fx), __fmul_rn(p[13], fy)), p[12]);
    const float tile = p[10];
    const float ux = __fsub_rn(gx, __fmul_rn(floorf(__fdiv_rn(gx, tile)), tile));
    const float uy = __fsub_rn(gy, __fmul_rn(floorf(__fdiv_rn(gy, tile)), tile));
    const float lim = __fmul_rn(p[9], tile);
    masked = ux < lim && uy < lim;
  }
  if (!masked) {
    // inverse rotation about the canvas centre, zero fill outside the 640x640 crop
    const float dx = __fsub_rn(fx, 319.5f), dy = __fsub_rn(fy, 319.5f);
    const float rx = __fadd_rn(__fadd_rn(__fmul_rn(p[6], dx), __fmul_rn(p[7], dy)), 319.5f);
    const float ry = __fadd_rn(__fsub_rn(__fmul_rn(p[6], dy), __fmul_rn(p[7], dx)), 319.5f);
    if (rx >= 0.f && rx <= FOV_CANVAS - 1.0f && ry >= 0.f && ry <= FOV_CANVAS - 1.0f) {
      // RandomResizedCrop window -> source pixel (bilinear, edges clamped)
      float sx = __fsub_rn(__fadd_rn(p[2], __fmul_rn(__fadd_rn(rx, 0.5f), __fdiv_rn(p[4], FOV_CANVAS))), 0.5f);
      float sy = __fsub_rn(__fadd_rn(p[3], __fmul_rn(__fadd_rn(ry, 0.5f), __fdiv_rn(p[5], FOV_CANVAS))), 0.5f);
      const float sw = p[1] - 1.0f, sh = p[0] - 1.0f;
      sx = fminf(fmaxf(sx, 0.f), sw);
      sy = fminf(fmaxf(sy, 0.f), sh);
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      const int x1 = x0 + 1 < (int)p[1] ? x0 + 1 : x0, y1 = y0 + 1 < (int)p[0] ? y0 + 1 : y0;
      const float ax = __fsub_rn(sx, (float)x0), ay = __fsub_rn(sy, (float)y0);
      const unsigned char* r0 = img + ((long long)y0 * W) * 3;
      const unsigned char* r1 = img + ((long long)y1 * W) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a = (float)r0[x0 * 3 + c], b = (float)r0[x1 * 3 + c];
        const float d = (float)r1[x0 * 3 + c], e = (float)r1[x1 * 3 + c];
        const float top = __fadd_rn(a, __fmul_rn(ax, __fsub_rn(b, a)));
        const float bot = __fadd_rn(d, __fmul_rn(ax, __fsub_rn(e, d)));
        v[c] = __fadd_rn(top, __fmul_rn(ay, __fsub_rn(bot, top)));
      }
    }
  }
  // additive noise (sum of four uniforms: exact integer hash, no transcendental functions)
  if (p[16] > 0.f || p[15] != 0.f) {
    const unsigned seed = (unsigned)p[17];
    const unsigned ix = (unsigned)(int)floorf(fx), iy = (unsigned)(int)floorf(fy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned h = fov_hash(seed, ix, iy, (unsigned)c + 4u * pix_id);
      const float u = (float)((h & 0xff) + ((h >> 8) & 0xff) + ((h >> 16) & 0xff) + (h >> 24));  // 0..1020, mean 510
      const float z = __fmul_rn(__fsub_rn(u, 510.0f), 0.0067929f);                               // ~N(0,1)
      v[c] = __fadd_rn(v[c], __fadd_rn(p[15], __fmul_rn(p[16], z)));
    }
  }
  // colour twist: v' = gain * (M v) + offset
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float m = __fadd_rn(__fadd_rn(__fmul_rn(p[18 + 3 * c], v[0]), __fmul_rn(p[19 + 3 * c], v[1])), __fmul_rn(p[20 + 3 * c], v[2]));
    rgb[c] = __fadd_rn(__fmul_rn(p[27], m), p[28]);
  }
}

__global__ __launch_bounds__(256) void foveate_kernel(const unsigned char* __restrict__ images, const float* __restrict__ params,
                                                      int B, int H, int W, int OS, unsigned char* __restrict__ out, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OS);
    long long t = i / OS;
    const int oy = (int)(t % OS);
    t /= OS;
    const int n = (int)(t % B);
    const int view = (int)(t / B);
    const float* p = params + (long long)n * 32;
    const unsigned char* img = images + (long long)n * H * W * 3;
    const float S = view == 0 ? 400.f : (view == 1 ? 240.f : (view == 2 ? 100.f : 30.f));
    const float ax = floorf(__fadd_rn(__fmul_rn(p[29], __fsub_rn(FOV_CANVAS, S)), 0.5f));
    const float ay = floorf(__fadd_rn(__fmul_rn(p[30], __fsub_rn(FOV_CANVAS, S)), 0.5f));
    const float s = __fdiv_rn(S, (float)OS);
    const int ns = view == 3 ? 1 : (view == 2 ? 3 : 4);   // sub-samples per axis of the resize footprint
    float acc[3] = {0.f, 0.f, 0.f};
    for (int ky = 0; ky < ns; ++ky)
      for (int kx = 0; kx < ns; ++kx) {
        const float cx = __fsub_rn(__fadd_rn(ax, __fmul_rn(__fadd_rn((float)ox, __fdiv_rn((float)kx + 0.5f, (float)ns)), s)), 0.5f);
        const float cy = __fsub_rn(__fadd_rn(ay, __fmul_rn(__fadd_rn((float)oy, __fdiv_rn((float)ky + 0.5f, (float)ns)), s)), 0.5f);
        float rgb[3];
        fov_canvas(img, H, W, p, cx, cy, (unsigned)(view * 16 + ky * 4 + kx), rgb);
        acc[0] = __fadd_rn(acc[0], rgb[0]);
        acc[1] = __fadd_rn(acc[1], rgb[1]);
        acc[2] = __fadd_rn(acc[2], rgb[2]);
      }
    const float inv = __fdiv_rn(1.0f, (float)(ns * ns));
    unsigned char* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float r = __fmul_rn(acc[c], inv);
      r = fminf(fmaxf(r, 0.f), 255.f);
      o[c] = (unsigned char)floorf(__fadd_rn(r, 0.5f));
    }
  }
}

extern "C" int maai_foveate_views_u8(const void* images, const float* params, int B, int H, int W, int OS, void* out,
                                     void* stream) {
  MAAI_CHECK_ARG(images && params && out && B > 0 && H > 0 && W > 0 && OS > 0, "foveate_views_u8: bad arguments");
  const long long total = 4ll * B * OS * OS;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(foveate_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const unsigned char*)images, params, B, H, W, OS, (unsigned char*)out, total);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
