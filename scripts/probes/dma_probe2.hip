// LDS-DMA staging pipeline alone (no MFMA): 3 workgroups per CU, each a 3-slot ring of 16 KB stages, counted vmcnt
// wait + barrier per stage as in conv_igemm.  Stage = RB bytes from each of 16384/RB rows of a [rows][512 B] operand
// (row blocks re-read by 8 workgroups like 8 column tiles).  Compares 64-byte and 128-byte row segments per stage.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int RB>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ x, int nblocks, int rowbytes, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPR = RB / 16, RPP = 256 / LPR, ROWS = 16384 / RB, NI = ROWS / RPP;  // NI = 4 either way
  const int tid = threadIdx.x, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int blk = blockIdx.x / 8;
  if (blk >= nblocks) return;
  const int r0 = tid / LPR, ch = tid % LPR;
  const char* base = x + ((long long)blk * ROWS + r0) * rowbytes + ch * 16;
  const int KT = rowbytes / RB;
  auto issue = [&](int kt) {
    char* dst = smem + (kt % 3) * 16384 + widu * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long long)i * RPP * rowbytes + kt * RB),
                                       (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
  };
  issue(0);
  if (KT > 1) issue(1);
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < KT) issue(kt + 2);
  }
  if (tid == 0 && smem[5] == 77) sink[0] = 1;
}

int main() {
  const int rowbytes = 512;
  const long long rows = 802816LL * 2;     // 0.82 GB operand
  char* x;
  int* sink;
  hipMalloc(&x, rows * rowbytes);
  hipMalloc(&sink, 4);
  hipMemset(x, 1, rows * rowbytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  for (int rb = 64; rb <= 128; rb *= 2) {
    const int nblocks = (int)(rows / (16384 / rb));
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (rb == 64)
        hipLaunchKernelGGL(probe<64>, dim3(nblocks * 8), dim3(256), 49152, 0, x, nblocks, rowbytes, sink);
      else
        hipLaunchKernelGGL(probe<128>, dim3(nblocks * 8), dim3(256), 49152, 0, x, nblocks, rowbytes, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)rows * rowbytes * 8;
      printf("row segment %3d B: %.3f ms  %.2f TB/s into LDS  (%.1f B/clk/CU at 2.1 GHz)\n", rb, ms, bytes / ms / 1e9,
             bytes / (ms * 1e-3) / 256 / 2.1e9);
    }
  }
  return 0;
}
