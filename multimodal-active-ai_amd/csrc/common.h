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
// Kernel-name notes for bench.py's per-kernel roofline rows (maai_kernel_names(1) switches them on): every launcher of a
// hot kernel calls MAAI_NOTE_KERNEL(kernel<...>) right before its launch; maai_last_kernel_name() then returns that
// instantiation's name as rocprofv3's kernel trace prints it (the demangled symbol, without "void" and the argument list).
extern "C" void maai_note_kernel(const void* host_function);
#define MAAI_NOTE_KERNEL(...) maai_note_kernel(reinterpret_cast<const void*>(&__VA_ARGS__))

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remember, per device, the largest
// size already granted (table: 64 ints, zero-initialised, one per kernel instantiation).
static inline void maai_ensure_lds(const void* fn, int lds, int* table) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& have = table[dev & 63];
  if (lds > have) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    have = lds;
  }
}

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

// a value as the storage type would hold it (bf16: one rounding; f32: unchanged)
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }

// Non-temporal 16-byte accesses for tensors that are streamed once (activations of gigabytes, far beyond the 256 MB
// Infinity Cache): measured on MI355X (scripts/probes/stream_probe.hip, 6.6 GB tensors) a 2-read-1-write stream moves
// 5.3 TB/s with plain accesses and 6.0 TB/s non-temporal, a 1-read-1-write stream 5.4 -> 6.4 TB/s.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16_nt(void* p, uint4 r) {
  const u32x4_t v = {r.x, r.y, r.z, r.w};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
}

// 16-byte vector of T unpacked to floats (8 bf16 or 4 f32)
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ void load_nt(const bf16_t* p) { raw = ld16_nt(p); }
  __device__ __forceinline__ void store_nt(bf16_t* p) const { st16_nt(p, raw); }
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
  __device__ __forceinline__ void load_nt(const float* p) {
    const uint4 v = ld16_nt(p);
    raw = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  __device__ __forceinline__ void store_nt(float* p) const {
    st16_nt(p, make_uint4(__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z), __float_as_uint(raw.w)));
  }
  __device__ __forceinline__ void zero() { raw = make_float4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float* f) const { f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w; }
  __device__ __forceinline__ void set(const float* f) { raw = make_float4(f[0], f[1], f[2], f[3]); }
};

// ---------------------------------------------------------------------------
// Normalise-on-load arithmetic (conv_igemm.h XF, conv_wgrad.hip XF): act(v*s + t [+ r(w*s2 + t2) | + w]) on one
// 16-byte chunk, with the fp32 operations and roundings of maai_bn_act_fwd / _fwd2, at the fewest vector
// instructions: the transform runs on the threads of a GEMM's K loop, where it competes with the MFMA issue.
// bf16: max(.,0) is taken AFTER the rounding, on the packed pairs, as a signed 16-bit max (bf16 is sign-magnitude:
// negative values and -0 are negative integers) — one v_pk_max_i16 per two elements instead of two v_max_f32 per
// element; identical for every non-NaN input.  The 1-bit mask (out > 0) of the non-negative result is min(half, 1)
// per 16-bit half (v_pk_min_u16) gathered with shifts.
// ---------------------------------------------------------------------------
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
  const s16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, v), z));
}
__device__ __forceinline__ uint32_t nonzero_bits_bf16x2(uint32_t v) {  // bit 0: low half != 0, bit 1: high half != 0
  const u16x2_t one = {1, 1};
  const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, v), one));
  return (m | (m >> 15)) & 3u;
}

template <typename T> struct XfMath;
template <> struct XfMath<bf16_t> {
  // v <- act(v*s + t (+ w'))  where w' = r(w*s2 + t2) if s2 else w;  returns the 1-bit mask if asked
  template <bool JOIN>
  __device__ static __forceinline__ unsigned run(Vec16<bf16_t>& v, const Vec16<bf16_t>& w, const float* s, const float* t,
                                                  const float* s2, const float* t2, int relu, bool want_bits) {
    float f[8];
    v.get(f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= s[e];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += t[e];
    if constexpr (JOIN) {
      float g[8];
      w.get(g);
      if (s2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = g[e] * s2[e] + t2[e];
        Vec16<bf16_t> r;
        r.set(g);
        r.get(g);  // the shortcut as a separate pass would have stored it
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += g[e];
    }
    v.set(f);
    if (relu) {
      v.raw.x = relu_bf16x2(v.raw.x);
      v.raw.y = relu_bf16x2(v.raw.y);
      v.raw.z = relu_bf16x2(v.raw.z);
      v.raw.w = relu_bf16x2(v.raw.w);
    }
    unsigned b = 0;
    if (want_bits) {
      if (relu) {
        b = nonzero_bits_bf16x2(v.raw.x) | (nonzero_bits_bf16x2(v.raw.y) << 2) | (nonzero_bits_bf16x2(v.raw.z) << 4) |
            (nonzero_bits_bf16x2(v.raw.w) << 6);
      } else {
        v.get(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
      }
    }
    return b;
  }
};
template <> struct XfMath<float> {
  template <bool JOIN>
  __device__ static __forceinline__ unsigned run(Vec16<float>& v, const Vec16<float>& w, const float* s, const float* t,
                                                  const float* s2, const float* t2, int relu, bool) {
    float f[4];
    v.get(f);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] *= s[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] += t[e];
    if constexpr (JOIN) {
      float g[4];
      w.get(g);
      if (s2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = g[e] * s2[e] + t2[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] += g[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    v.set(f);
    return 0u;
  }
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
