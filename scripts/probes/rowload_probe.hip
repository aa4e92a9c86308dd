// How fast can a workgroup-per-128-rows kernel pull a [M][256] bf16 tensor (512-byte rows) into registers, depending on the
// per-instruction access pattern?  The streaming kernels (conv_pws / conv_chain / conv_dfold) load the MFMA A layout directly:
// an instruction covers 16 rows x 64 bytes (lane = row + 16 * chunk).  Variants:
//   0  MFMA layout, non-temporal        (what the kernels do)
//   1  MFMA layout, plain loads
//   2  row-contiguous: an instruction covers 2 rows x 512 bytes (lane = 32 * row + chunk), non-temporal
//   3  row-contiguous, plain
//   4  MFMA layout but K-permuted so that an instruction covers 16 rows x 64 bytes of the SAME 128-byte line pair ... (= 0; kept for symmetry)
// Each workgroup (256 threads, 4 waves x 32 rows) reads its 128 rows x 512 B (+ a [M][64] side tensor, 128 B rows) and writes
// [M][64]; nothing else.   hipcc --offload-arch=gfx950 -O3 rowload_probe.hip -o rowload_probe && ./rowload_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld(const uint4* p, bool nt) {
  if (nt) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
__device__ __forceinline__ uint4 mix(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y + b.y, a.z ^ b.z, a.w + b.w); }

template <int VAR>
__global__ __launch_bounds__(256, 3) void probe(const uint4* __restrict__ g, const uint4* __restrict__ y2, uint4* __restrict__ out, long long M) {
  constexpr bool NT = (VAR == 0 || VAR == 2);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long long row0 = (long long)blockIdx.x * 128 + wid * 32;
  uint4 r[16];
  if (VAR <= 1) {
    // MFMA layout: i = 16-row group (2), kt = 8: lane -> row (lane & 15), chunk kt*4 + (lane >> 4)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        long long m = row0 + i * 16 + (lane & 15);
        if (m >= M) m = M - 1;
        r[i * 8 + kt] = ld(g + m * 32 + kt * 4 + (lane >> 4), NT);
      }
  } else {
    // row-contiguous: instruction j covers rows row0 + 2j, 2j+1 (32 chunks each)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      long long m = row0 + 2 * j + (lane >> 5);
      if (m >= M) m = M - 1;
      r[j] = ld(g + m * 32 + (lane & 31), NT);
    }
  }
  // the side tensor and the output in the row-store layout of the kernels' epilogues: 8 lanes per 128-byte row
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) acc = mix(acc, r[j]);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    long long m = row0 + it * 8 + (lane >> 3);
    if (m < M) {
      const uint4 v = y2[m * 8 + (lane & 7)];
      out[m * 8 + (lane & 7)] = mix(acc, v);
    }
  }
}

template <int VAR>
static void run(const uint4* g, const uint4* y2, uint4* out, long long M, const char* what) {
  const unsigned grid = (unsigned)((M + 127) / 128);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), 0, 0, g, y2, out, M);
  hipEventRecord(e0, 0);
  const int n = 5;
  for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), 0, 0, g, y2, out, M);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  const double bytes = (double)M * (512 + 128 + 128);
  printf("variant %d (%s): %.3f ms  %.2f TB/s\n", VAR, what, ms, bytes / ms / 1e9);
}

int main() {
  const long long M = 256ll * 224 * 224;
  uint4 *g, *y2, *out;
  hipMalloc(&g, M * 512); hipMalloc(&y2, M * 128); hipMalloc(&out, M * 128);
  hipMemset(g, 1, M * 512); hipMemset(y2, 2, M * 128);
  for (int pass = 0; pass < 2; ++pass) {
    run<0>(g, y2, out, M, "MFMA layout, nt");
    run<1>(g, y2, out, M, "MFMA layout, plain");
    run<2>(g, y2, out, M, "row-contiguous, nt");
    run<3>(g, y2, out, M, "row-contiguous, plain");
  }
  return 0;
}
