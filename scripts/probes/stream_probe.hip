// What does a BatchNorm-apply-shaped stream (2 reads + 1 write, or 1 read + 1 write, 16 B per lane) sustain on this
// device, and does the loop shape matter?  Variants: grid-stride one chunk per iteration (what bn.hip does), 4 chunks
// in flight per lane, non-temporal loads / stores, grid size.   hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntstore(uint4 r, uint4* p) {
  const u32x4 v = {r.x, r.y, r.z, r.w};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}

__device__ __forceinline__ uint4 op(uint4 a, uint4 b) {  // stands in for the per-element maths (cheap, not elidable)
  return make_uint4(a.x + (b.x & 0xffff0000u), a.y ^ b.y, a.z + b.z, a.w | b.w);
}

template <int NIN, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    uint4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      va[u] = NT ? ntload(a + i + u * stride) : a[i + u * stride];
      if (NIN == 2) vb[u] = NT ? ntload(b + i + u * stride) : b[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint4 r = NIN == 2 ? op(va[u], vb[u]) : op(va[u], va[u]);
      if (NT)
        ntstore(r, o + i + u * stride);
      else
        o[i + u * stride] = r;
    }
  }
  for (; i < n; i += stride) {
    const uint4 r = NIN == 2 ? op(a[i], b[i]) : op(a[i], a[i]);
    o[i] = r;
  }
}

// the real thing: bf16 BatchNorm apply + residual + ReLU (bn.hip), with the knobs under test
__device__ __forceinline__ void unpack(uint4 r, float* f) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
__device__ __forceinline__ unsigned pk(float a, float b) {
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
template <bool NT, bool HOIST, bool BITS>
__global__ __launch_bounds__(256) void bnpass(const uint4* __restrict__ y, const uint4* __restrict__ res, uint4* __restrict__ o,
                                              unsigned char* __restrict__ bits, const float* __restrict__ scale,
                                              const float* __restrict__ shift, long long n, int cpr) {
  float hs[8], ht[8];
  if (HOIST) {
    const int c = (int)(((long long)blockIdx.x * 256 + threadIdx.x) % cpr) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { hs[e] = scale[c + e]; ht[e] = shift[c + e]; }
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr) * 8;
    const uint4 vy = NT ? ntload(y + i) : y[i];
    const uint4 vr = NT ? ntload(res + i) : res[i];
    float f[8], g[8];
    unpack(vy, f);
    unpack(vr, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] *= HOIST ? hs[e] : scale[c + e];
      f[e] += HOIST ? ht[e] : shift[c + e];
      f[e] = fmaxf(f[e] + g[e], 0.f);
    }
    const uint4 r = make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    if (NT) ntstore(r, o + i); else o[i] = r;
    if (BITS) {
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
      bits[i] = (unsigned char)b;
    }
  }
}
template <bool NT, bool HOIST, bool BITS>
static void runbn(const char* name, const uint4* a, const uint4* b, uint4* o, unsigned char* bits, const float* sc, const float* sh,
                  long long n, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((bnpass<NT, HOIST, BITS>), dim3(grid), dim3(256), 0, 0, a, b, o, bits, sc, sh, n, 32);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s grid %6d: %.3f ms  %.2f TB/s\n", name, grid, best, (double)n * 48 / best / 1e9);
}

template <int NIN, int UNROLL, bool NT>
static void run(const char* name, const uint4* a, const uint4* b, uint4* o, long long n, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<NIN, UNROLL, NT>), dim3(grid), dim3(256), 0, 0, a, b, o, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)n * 16 * (NIN + 1);
  printf("%-34s grid %6d: %.3f ms  %.2f TB/s\n", name, grid, best, bytes / best / 1e9);
}

int main() {
  const long long n = 12845056LL * 256 * 2 / 16;  // one 256-channel bf16 tensor at 224^2 x 256 images: 6.6 GB
  uint4 *a, *b, *o;
  hipMalloc(&a, n * 16);
  hipMalloc(&b, n * 16);
  hipMalloc(&o, n * 16);
  hipMemset(a, 1, n * 16);
  hipMemset(b, 2, n * 16);
  unsigned char* bits;
  float *sc, *sh;
  hipMalloc(&bits, n);
  hipMalloc(&sc, 1024);
  hipMalloc(&sh, 1024);
  hipMemset(sc, 0, 1024);
  hipMemset(sh, 0, 1024);
  for (int grid : {8192, 32768, 131072}) {
    runbn<false, false, false>("bn pass plain", a, b, o, bits, sc, sh, n, grid);
    runbn<false, false, true>("bn pass plain + bits", a, b, o, bits, sc, sh, n, grid);
    runbn<true, false, false>("bn pass nt", a, b, o, bits, sc, sh, n, grid);
    runbn<true, true, false>("bn pass nt hoist", a, b, o, bits, sc, sh, n, grid);
    runbn<true, true, true>("bn pass nt hoist + bits", a, b, o, bits, sc, sh, n, grid);
    runbn<false, true, true>("bn pass hoist + bits", a, b, o, bits, sc, sh, n, grid);
  }
  for (int grid : {8192, 32768, 131072}) {
    run<2, 1, false>("2 in 1 out, 1 chunk/iter", a, b, o, n, grid);
    run<2, 4, false>("2 in 1 out, 4 chunks in flight", a, b, o, n, grid);
    run<2, 4, true>("2 in 1 out, 4 chunks, non-temporal", a, b, o, n, grid);
    run<2, 1, true>("2 in 1 out, 1 chunk, non-temporal", a, b, o, n, grid);
    run<1, 1, false>("1 in 1 out, 1 chunk/iter", a, b, o, n, grid);
    run<1, 4, false>("1 in 1 out, 4 chunks in flight", a, b, o, n, grid);
    run<1, 4, true>("1 in 1 out, 4 chunks, non-temporal", a, b, o, n, grid);
  }
  return 0;
}
