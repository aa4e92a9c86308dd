// Backward of a 64 -> 256 bottleneck's last unit (conv3 + bn3, resnet.py:118-119) in ONE persistent launch, bf16:
//     dz3 = k1*g - k2 - k3*y3                 BatchNorm-backward apply (g = gradient of bn3's output, masked upstream)
//     dx  = (dz3 . W3) * [a2 > 0]             data gradient into conv3's input a2 = relu(bn2(y2)), + bn2's backward sums
//     dW3 += dz3^T . a2                       weight gradient
// The unfused sequence (apply-on-load data gradient + weight gradient) writes dz3 (6.6 GB at 224^2 x 256 images) only for
// the weight gradient to read it back; here g and y3 are read once and dz3 exists only in registers and LDS.
//
// Persistent workgroups (two per CU, 4 waves, 64 pixel rows per tile, tiles strided over the grid).  conv3's weights (32 KB,
// data-gradient form) stay in LDS for the whole launch; the weight gradient is accumulated in registers (wave w owns output
// channels 64w .. 64w+63 of dW3: 64 fp32 registers) and flushed with fp32 atomics once at the end.  Per tile: g, y3 (A layout
// of the MFMA: a lane holds 8 channels of one pixel) and y2 arrive in registers; dz3 is formed in place, written pixel-major
// to LDS (the weight gradient reads it transposed, ds_read_b64_tr_b16) and multiplied from registers by the resident weights
// (no barrier in that loop); the epilogue masks by y2*s2 + t2 > 0, stores dx, adds bn2's sums and drops a2 = relu(r(y2*s2 +
// t2)) into LDS for the weight gradient.  All vector-memory traffic is compiler-visible loads / stores: no manual vmcnt.
// Arithmetic of dz3, dx and the sums as in conv_igemm.h AXF + EMODE 6 (same roundings): dx is bit-identical to that path.
#include "conv_igemm.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

struct Bwd3Args {
  const void* g;     // [M][256]
  const void* y3;    // [M][256]
  const void* y2;    // [M][64]
  const void* wd;    // [64][256]: conv3's weights in data-gradient form (rows = input channels of conv3)
  const float* k1;   // [256]
  const float* k2;
  const float* k3;
  const float* mean2;  // [64]
  const float* s2;
  const float* t2;
  void* dx;          // [M][64]
  int accumulate;    // dx += (the shortcut branch of a projection block adds to the main branch's gradient, in place)
  float* slab;       // [gridDim][2][64]
  float* dw;         // [256][64] fp32, += (zeroed by the caller)
  long long M;
  int ntiles;
};

// Two workgroups per CU overlap each other's phases (a one-workgroup variant that prefetched the next tile's operands into a
// second register set measured 4.7 vs 3.75 ms).  64-row tiles, 16 rows per wave: the wave's C staging area IS its 16 rows of
// the a2 image — each row is read back as chunks before the same lanes overwrite it with a2 — which keeps LDS under 80 KB.
// ACC: dx += (in place).
template <bool ACC>
__global__ __launch_bounds__(256, 2) void conv_bwd3_kernel(Bwd3Args a) {
  typedef bf16_t T;
  constexpr int TM = 1;
  constexpr int KC = 256, NC = 64, KT = KC / 32, TN = NC / 16, LDC = NC, CW = 16 * LDC * 2, RB = 128, BM = 64 * TM;
  constexpr int WSM = KT * 4096, DZT = 4 * BM * RB, A2T = BM * RB;
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wsm = smem;                  // [kt][64 rows][64 B], swizzled: the resident weights
  char* dzt = smem + WSM;            // four [BM rows][64 ch] sub-tiles of dz3 (128-byte rows, swizzled for transposed reads)
  char* a2t = dzt + DZT;             // [BM rows][64 ch]
  float* coef = reinterpret_cast<float*>(a2t + A2T);     // k1 | k2 | k3 (256 each)
  char* cws = a2t;                   // wave-private C areas (see above)
  float* red = reinterpret_cast<float*>(dzt);            // [4 waves][2][64]: after the last tile

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int li = lane & 15, gl = lane >> 4, er = lane >> 3, ec = lane & 7;
  const T* __restrict__ gp = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ y3p = reinterpret_cast<const T*>(a.y3);
  const T* __restrict__ y2p = reinterpret_cast<const T*>(a.y2);
  const T* __restrict__ wd = reinterpret_cast<const T*>(a.wd);
  T* __restrict__ dx = reinterpret_cast<T*>(a.dx);

  // ---- once per workgroup: weights and coefficient tables into LDS ----
  {
    const int r = tid >> 2, c = tid & 3;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const uint4 v = *reinterpret_cast<const uint4*>(wd + (long long)r * KC + kt * 32 + c * 8);
      *reinterpret_cast<uint4*>(wsm + kt * 4096 + r * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4)) = v;
    }
    for (int i = tid; i < 3 * KC; i += 256) coef[i] = i < KC ? a.k1[i] : (i < 2 * KC ? a.k2[i - KC] : a.k3[i - 2 * KC]);
  }
  float m2[8], sc2[8], sh2[8], s1[8], s2[8];   // this lane's epilogue channels ec*8 .. +8: mean | scale | shift of bn2; its sums
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    m2[e] = a.mean2[ec * 8 + e];
    sc2[e] = a.s2[ec * 8 + e];
    sh2[e] = a.t2[ec * 8 + e];
    s1[e] = 0.f;
    s2[e] = 0.f;
  }
  f32x4 dwacc[4][TN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < TN; ++j) dwacc[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment offsets
  const int foff = li * 64 + ((gl ^ (((li >> 3) & 1) << 1)) << 4);            // resident weights (rows = ci)
  const int q = li >> 2, p = li & 3, r0 = 4 * gl + q;                          // transposed reads: this lane's row in a 16-row half
  int offA[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) offA[c] = r0 * RB + (((2 * c + (p >> 1)) ^ (((r0 >> 1) & 3) << 1)) << 4) + ((p & 1) << 3);
  const int baseB = r0 * RB + ((p & 1) << 3);
  const int swzB = ((p >> 1) ^ (((r0 >> 1) & 3) << 1)) << 4;
  char* cw = cws + widu * CW;
  const uint32_t cwa = (uint32_t)(uintptr_t)(cw + ((gl * 4) * LDC + li) * 2);

  auto load_tile = [&](int t, uint4 (&G)[TM][KT], uint4 (&Y)[TM][KT], uint4 (&Y2)[TM][2], uint4 (&P)[ACC ? TM : 1][2]) {
    const long long row0 = (long long)t * BM + widu * (16 * TM);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      long long m = row0 + i * 16 + li;
      if (m >= a.M) m = a.M - 1;
      const T* sg = gp + m * KC + gl * 8;
      const T* sy = y3p + m * KC + gl * 8;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        G[i][kt] = ld16_nt(sg + kt * 32);
        Y[i][kt] = ld16_nt(sy + kt * 32);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        long long m2r = row0 + i * 16 + er + 8 * it;
        if (m2r >= a.M) m2r = a.M - 1;
        Y2[i][it] = *reinterpret_cast<const uint4*>(y2p + m2r * NC + ec * 8);
        if constexpr (ACC) P[i][it] = *reinterpret_cast<const uint4*>(dx + m2r * NC + ec * 8);
      }
    }
  };

  auto process = [&](int t, uint4 (&G)[TM][KT], uint4 (&Y)[TM][KT], uint4 (&Y2)[TM][2], uint4 (&P)[ACC ? TM : 1][2]) {
    const long long row0 = (long long)t * BM + widu * (16 * TM);
    // everyone has finished the weight-gradient reads of the previous tile's LDS images (and, first tile, the tables are in)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- dz3 = k1*g - k2 - k3*y3, in place; pixel-major copy to LDS ----
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float q1[8], q2[8], q3[8];
      const float* cs = coef + kt * 32 + gl * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q1[e] = cs[e];
        q2[e] = cs[KC + e];
        q3[e] = cs[2 * KC + e];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        Vec16<T> v, w2;
        v.raw = G[i][kt];
        w2.raw = Y[i][kt];
        float d[8], yy[8];
        v.get(d);
        w2.get(yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = q1[e] * d[e] - q2[e] - q3[e] * yy[e];
        v.set(d);
        if (row0 + i * 16 + li >= a.M) v.zero();
        G[i][kt] = v.raw;
        const int row = widu * (16 * TM) + i * 16 + li;
        *reinterpret_cast<uint4*>(dzt + (kt >> 1) * (BM * RB) + row * RB + ((((kt & 1) * 4 + gl) ^ (((row >> 1) & 3) << 1)) << 4)) = v.raw;
      }
    }
    // ---- data gradient: [32 rows][256] . [256][64], weights resident ----
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      frag_t bfr[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(wsm + kt * 4096 + foff + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const frag_t af = __builtin_bit_cast(frag_t, G[i][kt]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af, bfr[j], acc[i][j]);
      }
    }
    // ---- epilogue: mask by y2*s2 + t2 > 0, store, bn2's sums; a2 into LDS for the weight gradient ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 v = acc[i][j];
        const uint32_t p01 = pack_bf16x2(v[0], v[1]);
        const uint32_t p23 = pack_bf16x2(v[2], v[3]);
        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p01), "n"(j * 32),
                     "n"(j * 32 + LDC * 2)
                     : "memory");
        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p23),
                     "n"(j * 32 + LDC * 4), "n"(j * 32 + LDC * 6)
                     : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const long long m = row0 + i * 16 + er + 8 * it;
        const bool ok = m < a.M;
        Vec16<T> v, vy;
        v.load(reinterpret_cast<const T*>(cw) + (er + 8 * it) * LDC + ec * 8);
        vy.raw = Y2[i][it];
        float fv[8], fy[8];
        v.get(fv);
        vy.get(fy);
        if constexpr (ACC) {
          Vec16<T> vp;
          vp.raw = P[i][it];
          float fo[8];
          vp.get(fo);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fo[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = (fy[e] * sc2[e] + sh2[e]) > 0.f ? fv[e] : 0.f;
        v.set(fv);
        v.get(fv);   // the rounded value being stored is what a separate reduction pass would read back
        if (ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += fv[e];
            s2[e] += fv[e] * (fy[e] - m2[e]);
          }
          v.store(dx + m * NC + ec * 8);
        }
        // a2 = relu(r(y2*s2 + t2)) (maai_bn_act_fwd arithmetic); rows past the end contribute nothing
        Vec16<T> w2;
        XfMath<T>::template run<false>(vy, w2, sc2, sh2, nullptr, nullptr, 1, false);
        if (!ok) vy.zero();
        const int row = widu * (16 * TM) + i * 16 + er + 8 * it;
        *reinterpret_cast<uint4*>(a2t + row * RB + ((ec ^ (((row >> 1) & 3) << 1)) << 4)) = vy.raw;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's rows of dz3 and a2 are in LDS
    // ---- weight gradient: dW3[64w .. +64][64] += dz3^T . a2 over the tile's rows ----
#pragma unroll
    for (int s = 0; s < 2 * TM; ++s) {
      bf16x8 af[4], bfr[TN];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char* base = dzt + widu * (BM * RB) + s * 32 * RB + offA[c];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(base + 16 * RB));
        af[c] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* base = a2t + s * 32 * RB + baseB + (swzB ^ (j << 5));
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(base + 16 * RB));
        bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j) dwacc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfr[j], dwacc[c][j], 0, 0, 0);
    }
  };

  // ---- tiles strided over the grid ----
  const int G = gridDim.x;
  {
    uint4 GA[TM][KT], YA[TM][KT], Y2A[TM][2], PA[ACC ? TM : 1][2];
    for (int t = blockIdx.x; t < a.ntiles; t += G) {
      load_tile(t, GA, YA, Y2A, PA);
      process(t, GA, YA, Y2A, PA);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();   // everyone is done with the LDS images: the sums' scratch reuses them

  // ---- bn2's sums: one slab row per workgroup ----
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      s1[e] += __shfl_xor(s1[e], o);
      s2[e] += __shfl_xor(s2[e], o);
    }
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(widu * 2 + 0) * NC + lane * 8 + e] = s1[e];
      red[(widu * 2 + 1) * NC + lane * 8 + e] = s2[e];
    }
  }
  __syncthreads();
  if (tid < 2 * NC) {
    const int which = tid / NC, c = tid - which * NC;
    a.slab[((long long)blockIdx.x * 2 + which) * NC + c] =
        red[which * NC + c] + red[(2 + which) * NC + c] + red[(4 + which) * NC + c] + red[(6 + which) * NC + c];
  }
  // ---- weight gradient: C layout row (co) = 4*gl + r, column (ci) = li ----
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = widu * 64 + c * 16 + 4 * gl + r;
        const int ci = j * 16 + li;
        atomicAdd(a.dw + (long long)co * NC + ci, dwacc[c][j][r]);
      }
}

// g, y3: [M][256]; y2: [M][64]; wd: conv3's weights as [64][256] (data-gradient form); k1..k3 [256]; mean2, s2, t2 [64];
// dx [M][64] (accumulate != 0: += in place, the sums are those of the stored result); slab [*slab_rows][2][64] (rows = workgroups launched, returned by maai_conv_bwd3_rows); dw [256][64] fp32 +=.
extern "C" int maai_conv_bwd3_rows(long long M) {
  const long long tiles = (M + 63) / 64;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long wgs = 2LL * cus;
  return (int)(tiles < wgs ? tiles : wgs);
}

extern "C" int maai_conv_bwd3(const void* g, const void* y3, const void* y2, const void* wd, const float* k1, const float* k2,
                              const float* k3, const float* mean2, const float* s2, const float* t2, void* dx, float* slab,
                              float* dw, long long M, int accumulate, void* stream) {
  MAAI_CHECK_ARG(g && y3 && y2 && wd && k1 && k2 && k3 && mean2 && s2 && t2 && dx && slab && dw && M > 0, "conv_bwd3: null pointer");
  MAAI_CHECK_ARG(M < (1ll << 31), "conv_bwd3: pixel count must fit 31 bits");
  constexpr int BM = 64;
  Bwd3Args a;
  a.g = g; a.y3 = y3; a.y2 = y2; a.wd = wd; a.k1 = k1; a.k2 = k2; a.k3 = k3; a.mean2 = mean2; a.s2 = s2; a.t2 = t2;
  a.dx = dx; a.accumulate = accumulate; a.slab = slab; a.dw = dw; a.M = M; a.ntiles = (int)((M + BM - 1) / BM);
  constexpr int lds = 8 * 4096 + 4 * BM * 128 + BM * 128 + 3 * 256 * 4;
  const int grid = maai_conv_bwd3_rows(M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static int attr[2][64] = {{0}};
  if (accumulate) {
    maai_ensure_lds(reinterpret_cast<const void*>(&conv_bwd3_kernel<true>), lds, attr[0]);
    MAAI_NOTE_KERNEL(conv_bwd3_kernel<true>);
    hipLaunchKernelGGL((conv_bwd3_kernel<true>), dim3((unsigned)grid), dim3(256), lds, st, a);
  } else {
    maai_ensure_lds(reinterpret_cast<const void*>(&conv_bwd3_kernel<false>), lds, attr[1]);
    MAAI_NOTE_KERNEL(conv_bwd3_kernel<false>);
    hipLaunchKernelGGL((conv_bwd3_kernel<false>), dim3((unsigned)grid), dim3(256), lds, st, a);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
