// Shared device/host helpers for the MI355X (gfx950, CDNA4) SimCLR hot path.
// wave = 64 lanes; MFMA fragment maps follow cdna_hip_programming.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA bf16 operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MAAI_OK 0
#define MAAI_ERR_ARG 1
#define MAAI_ERR_LAUNCH 2
#define MAAI_ERR_UNSUPPORTED 3

extern "C" void maai_set_error(const char* msg);

// also clears any stale (sticky) runtime error so that MAAI_CHECK_LAUNCH reports only our own launch
#define MAAI_CHECK_ARG(cond, msg)        \
  do {                                   \
    (void)hipGetLastError();             \
    if (!(cond)) {                       \
      maai_set_error(msg);               \
      return MAAI_ERR_ARG;               \
    }                                    \
  } while (0)

#define MAAI_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) {                                  \
      maai_set_error(hipGetErrorString(e__));                 \
      return MAAI_ERR_LAUNCH;                                 \
    }                                                         \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN-preserving: a plain __bf16 cast, which hipcc lowers to v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_hw v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// value of the lane with index lane^1 (DPP quad_perm [1,0,3,2]; no LDS, no wait)
__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// storage-type traits: T = bf16_t or float
template <typename T> struct Store;
template <> struct Store<bf16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Store<float> {
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};

// 16-byte vector of T unpacked to floats (8 bf16 or 4 f32)
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float* f) const {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void set(const float* f) {
    raw.x = pack_bf16x2(f[0], f[1]);
    raw.y = pack_bf16x2(f[2], f[3]);
    raw.z = pack_bf16x2(f[4], f[5]);
    raw.w = pack_bf16x2(f[6], f[7]);
  }
};
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
  __device__ __forceinline__ void zero() { raw = make_float4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float* f) const { f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w; }
  __device__ __forceinline__ void set(const float* f) { raw = make_float4(f[0], f[1], f[2], f[3]); }
};

// XCD-aware bijective block remap (cdna_hip_programming.md §5 "XCD swizzle must be
// bijective"): blocks b and b+8 share an XCD, so give each XCD a contiguous
// range of logical tiles; speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
