// Persistent, software-pipelined pointwise convolution (forward 1x1 stride-1 layers with BatchNorm statistics,
// resnet.py:28 conv1x1 + the statistics of resnet.py:106-133), bf16.
//
// Why: the one-tile-per-workgroup kernel (conv_fwd.hip) runs each 128x128 tile as a serial chain — first-stage load
// latency, K-steps separated by barriers, LDS transpose, stores — and with three tiles resident per CU the matrix
// pipe of the mid-network layers (K = 128..512 for 4K output columns) sits at 23 % (profiles/r01_pmc_pw256x1024_sq.json),
// with no instruction count or occupancy knob moving it.  Here two workgroups per CU stay resident and walk tiles:
// the LDS-DMA ring is indexed by a stage counter that runs ACROSS tiles, so while a tile's epilogue drains through
// its own LDS region the first K-steps of the next tile are already streaming in.
//
// vmcnt: LDS-DMA loads and the epilogue's global stores share the counter and retire in issue order on gfx9-family
// parts, so "stage g has landed" is a counted wait on everything issued after it: the younger ring stages plus, for
// the two iterations that follow an epilogue, that epilogue's store instructions (exactly 8 + 1 per wave on full
// tiles; partial tiles count none, which only makes the wait stricter).  No __syncthreads() in the loop: it would
// drain the ring (s_waitcnt vmcnt(0)); raw s_barrier + explicit lgkmcnt waits instead.
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is set by hand for the LDS-DMA instructions below
#include "common.h"
#include "conv_pp.h"
#include "maai_internal.h"

__device__ uint4 g_ppzero64[4];

// LDS-DMA as inline assembly: issued through the builtin, the compiler tracks the LDS write and puts its own
// s_waitcnt vmcnt(0) in front of LDS reads it cannot prove disjoint — which here would drain the cross-tile ring and
// serialise the epilogue's stores.  All ordering in this kernel is by the explicit counted waits below.
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst) {
  const unsigned ldsu = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(ldsu) : "memory", "m0");
}

template <int NL, int S>
__device__ __forceinline__ void wait_stage(int ahead, bool stores_pending) {
  // at most `ahead` younger stages (NL instructions each) and, if pending, the S epilogue stores may stay in flight
  if (ahead >= 1) {
    if (stores_pending)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + S) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
  } else {
    if (stores_pending)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

__global__ __launch_bounds__(256, 2) void pw_persist_kernel(PpArgs a) {
  constexpr int BM = 128, BN = 128, BK = 32, NSTAGE = 3;
  constexpr int STAGE = (BM + BN) * 64;        // 16 KB per ring slot
  constexpr int LDC = BN + 8;                  // C-tile row pitch (elements)
  constexpr int CROWS = 64;                    // the C tile drains in two 64-row phases
  constexpr int NL = 4;                        // LDS-DMA instructions per thread per stage (2 A + 2 B)
  constexpr int SST = 9;                       // store instructions per wave per full-tile epilogue (8 rows groups + statistics)
  typedef bf16x8 frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  bf16_t* ctile = reinterpret_cast<bf16_t*>(smem + NSTAGE * STAGE);
  float* red = reinterpret_cast<float*>(smem + NSTAGE * STAGE + CROWS * LDC * 2);  // [8][2][BN]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int wm = wid >> 1, wn = wid & 1;
  const int r0 = tid >> 2;
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ w = a.w;
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_ppzero64);
  const int KT = a.Cin / BK;

  // tiles of this workgroup: its XCD (blockIdx % 8) owns a contiguous range of tiles, column tile fastest, so the
  // workgroups that share an A tile sit on one XCD's L2 at about the same time
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
  const int per = (a.ntiles + 7) / 8;
  const int t_lo = xcd * per;
  const int t_hi = (t_lo + per < a.ntiles) ? t_lo + per : a.ntiles;
  int t_first = t_lo + jx;
  if (t_first >= t_hi) return;
  const int mytiles = (t_hi - t_first + nx - 1) / nx;
  const int total = mytiles * KT;

  // ---- issue side ----
  long long ab[2];
  bool aok[2];
  const bf16_t* wp[2];
  int i_tile = t_first, i_kt = 0, issued = 0;
  auto set_issue_tile = [&](int t) {
    const int mb = t / a.nNB, nb = t - mb * a.nNB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long m = (long long)mb * BM + r0 + 64 * i;
      aok[i] = m < a.M;
      ab[i] = m * a.Cin + chunk * 8;
      wp[i] = w + (long long)(nb * BN + r0 + 64 * i) * a.Cin + chunk * 8;
    }
  };
  auto issue = [&]() {
    char* sa = ring + (issued % NSTAGE) * STAGE + widu * 1024;
    char* sb = sa + BM * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* src = aok[i] ? x + ab[i] + i_kt * BK : zsrc;
      dma16(src, sa + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(wp[i] + i_kt * BK, sb + i * 4096);
    ++issued;
    if (++i_kt == KT) {
      i_kt = 0;
      i_tile += nx;
      if (i_tile < t_hi) set_issue_tile(i_tile);
    }
  };
  set_issue_tile(t_first);
  for (int s = 0; s < NSTAGE - 1 && s < total; ++s) issue();

  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) << 1)) << 4);
  int g = 0;              // stage being multiplied
  int since_epi = 2;      // iterations since the last epilogue's stores were issued (>= 2: they have retired in order)
  bool epi_full = false;  // that epilogue issued exactly SST store instructions per wave

  for (int t = t_first; t < t_hi; t += nx) {
    const int mb = t / a.nNB, nb = t - mb * a.nNB;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < KT; ++kt, ++g) {
      wait_stage<NL, SST>(issued - g - 1, since_epi < 2 && epi_full);
      if (since_epi < 2 && !epi_full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (partial tile: unknown store count)
      __builtin_amdgcn_s_barrier();
      if (issued < total) issue();  // refills the slot of stage g-1, which everyone has finished reading
      ++since_epi;
      const char* sa = ring + (g % NSTAGE) * STAGE + (wm * 64) * 64 + foff;
      const char* sb = ring + (g % NSTAGE) * STAGE + BM * 64 + (wn * 64) * 64 + foff;
      frag_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const frag_t*>(sa + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(sb + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (its own LDS: the ring keeps filling) ----
    if (a.stats) {
      float* sred = red + ((wm * 4 + (lane >> 4)) * 2) * BN + wn * 64 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 lo = {acc[i][j][0], acc[i][j][1]}, hi = {acc[i][j][2], acc[i][j][3]};
          s2 += lo;
          q2 = __builtin_elementwise_fma(lo, lo, q2);
          s2 += hi;
          q2 = __builtin_elementwise_fma(hi, hi, q2);
        }
        sred[j * 16] = s2.x + s2.y;
        sred[BN + j * 16] = q2.x + q2.y;
      }
    }
    const bool full = (long long)(mb + 1) * BM <= a.M;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (wm == ph) {
        const uint32_t ca = (uint32_t)(uintptr_t)(ctile + ((lane >> 4) * 4) * LDC + wn * 64 + (lane & 15));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc[i][j];
            const uint32_t p01 = pack_bf16x2(v[0], v[1]);
            const uint32_t p23 = pack_bf16x2(v[2], v[3]);
            asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(ca), "v"(p01),
                         "n"((i * 16 * LDC + j * 16) * 2), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 2)
                         : "memory");
            asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(ca), "v"(p23),
                         "n"((i * 16 * LDC + j * 16) * 2 + LDC * 4), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 6)
                         : "memory");
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (ph == 0 && a.stats) {  // 256 threads = 2 x BN sums
        const int which = tid >> 7, c = tid & 127;
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tsum += red[(k * 2 + which) * BN + c];
        a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] = tsum;
      }
      const long long row_base = (long long)mb * BM + ph * CROWS;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = (tid >> 4) + it * 16, ch = tid & 15;
        if (full || row_base + row < a.M) {
          const uint4 v = *reinterpret_cast<const uint4*>(ctile + row * LDC + ch * 8);
          *reinterpret_cast<uint4*>(a.y + (row_base + row) * a.Cout + nb * BN + ch * 8) = v;
        }
      }
      if (ph == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // phase 0's reads of the C tile are done
        __builtin_amdgcn_s_barrier();
      }
    }
    since_epi = 0;
    epi_full = full && a.stats != nullptr;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int maai_pp_conv_launch(PpArgs a, hipStream_t st) {
  constexpr int lds = 3 * 16384 + 64 * 136 * 2 + 16 * 128 * 4;  // ring + C tile + statistics scratch = 74,752 B
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  static int ncu = 0;
  if (ncu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
  }
  int grid = 2 * ncu;  // two resident workgroups per CU
  grid = (grid + 7) / 8 * 8;
  hipLaunchKernelGGL(pw_persist_kernel, dim3(grid), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
