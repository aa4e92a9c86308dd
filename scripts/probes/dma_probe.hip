// L2 -> LDS staging rate of global_load_lds_dwordx4 for two row widths per stage (64 B vs 128 B contiguous per row),
// in the access pattern of the pointwise implicit GEMM: 128-row tiles of a [M][512 B] operand, each tile read by 8
// workgroups (the 8 column tiles), 3 workgroups resident per CU.  Build: hipcc --offload-arch=gfx950 -O3 dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int RB>  // bytes per row per stage: 64 or 128
__global__ __launch_bounds__(256) void probe(const char* __restrict__ x, int ntiles, int rowbytes, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPR = RB / 16;           // lanes per row
  constexpr int ROWS = 256 / LPR;        // rows per 256-thread pass
  constexpr int NI = 128 / ROWS;         // passes per 128-row tile
  const int tid = threadIdx.x, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int tile = blockIdx.x / 8;
  if (tile >= ntiles) return;
  const int r0 = tid / LPR, ch = tid % LPR;
  const char* base = x + ((long long)tile * 128 + r0) * rowbytes + ch * 16;
  const int KT = rowbytes / RB;
  for (int kt = 0; kt < KT; ++kt) {
    char* dst = smem + (kt & 1) * (128 * RB) + widu * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long long)i * ROWS * rowbytes + kt * RB),
                                       (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
    if (kt & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[5] == 77) sink[0] = 1;
}

int main() {
  const int rowbytes = 512;                // Cin = 256 bf16
  const long long M = 802816;              // 256 x 56 x 56
  const int ntiles = (int)(M / 128);
  char* x;
  int* sink;
  hipMalloc(&x, M * rowbytes);
  hipMalloc(&sink, 4);
  hipMemset(x, 1, M * rowbytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rb = 64; rb <= 128; rb *= 2) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (rb == 64)
        hipLaunchKernelGGL(probe<64>, dim3(ntiles * 8), dim3(256), 2 * 128 * 64, 0, x, ntiles, rowbytes, sink);
      else
        hipLaunchKernelGGL(probe<128>, dim3(ntiles * 8), dim3(256), 2 * 128 * 128, 0, x, ntiles, rowbytes, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)M * rowbytes * 8;
      printf("row bytes per stage %3d: %.3f ms  %.2f TB/s into LDS  (%.1f B/clk/CU at 2.4 GHz)\n", rb, ms, bytes / ms / 1e9,
             bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
