// Data gradient of a FOLDED 64 -> 256 unit (conv3 + bn3 of layer 1's bottlenecks and layer1.0's projection shortcut,
// resnet.py:118-123; engine._FOLD, csrc/fold.hip), bf16, one launch:
//     dx = ( [g | a2] . Wcat^T + cn + dg*a2 ) * [a2 > 0]  Wcat[k] = [ k1*W[:,k] | -(W^T diag(k3) W)[:,k], diagonal in dg ]   (64 x 320)
// where g [M][256] is the gradient of bn3's OUTPUT as the block above left it (no BatchNorm-backward apply pass, no dz3
// tensor, y3 not read), a2 = relu(bn2(y2)) is formed on load from the raw y2 [M][64], + the BatchNorm-backward partial sums
// of the unit below (sum dx, sum dx*(y2 - mean2)) and optionally dx += (the shortcut branch accumulates onto the main one).
// It replaces conv_bwd3.hip's launch where the unit is folded: 9.9 GB instead of 16.4 GB per launch at 224^2 x 256 images.
//
// Structure of conv_pws.hip (streaming kernel): a workgroup owns 128 pixel rows, a wave 32 of them; the WHOLE K extent of
// the wave's rows — 8 chunks of g, 2 of y2 — goes straight into registers in MFMA A layout (every byte in flight at once,
// read once), the 40 KB of weights stream through an LDS-DMA ring (L2-resident), one column tile of 64 output channels.
// Epilogue as EMODE 6 of conv_igemm.h on the wave's private C area: y2 (and the previous dx) in the row-store layout were
// requested before the K loop.
#include "conv_igemm.h"
#include <stdlib.h>

struct DfoldArgs {
  const void* g;       // [M][256]
  const void* y2;      // [M][64]
  const void* w;       // [64][320]: Wcat (maai_fold_dgrad_w with pitch 320)
  const float* cn;     // [64]
  const float* dg;     // [64] fp32: the diagonal of -(W^T diag(k3) W), kept out of Wcat (nullable)
  const float* mean2;  // [64] bn2: mean | scale | shift
  const float* s2;
  const float* t2;
  void* dx;            // [M][64]
  float* slab;         // [nMB][2][64]
  long long M;
  int nMB;
};

template <int N>
__device__ __forceinline__ void dfold_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");   // (lgkmcnt(0): the ring invariant, as in conv_pws.hip)
}

// RESIDENT: all ten weight stages (40 KB) are requested at once, right behind the operand loads, and the K loop runs behind ONE
// wait + barrier instead of ten (the ring version waits for an L2 round trip and a barrier per 8 MFMAs of a wave).
// Three workgroups per CU (12 waves, 168 registers each).  A workgroup's life is a latency chain — operand loads, weights,
// K loop, epilogue, ~12 us of which ~2 us depend on its bytes (halving the rows per workgroup changed it from 12.4 to 11.5 us) —
// so what a CU streams is (workgroups resident) x (bytes each) / (that chain).  At three the first build spilled 24-42
// registers per lane (25-45 % extra traffic on a kernel that has nothing but traffic); the 40 registers in question were the
// epilogue's per-channel coefficients, held through the K loop: they are read from the LDS coefficient table in the epilogue
// now.  2.45 -> 2.11 ms per launch (4.0 -> 4.7 TB/s), accumulate form 2.64 -> 2.51.
template <bool ACC, bool RESIDENT>
__global__ __launch_bounds__(256, 3) void conv_dfold_kernel(DfoldArgs a) {
  typedef bf16_t T;
  constexpr int TM = 2, BM = 128, KC1 = 256, KC2 = 64, K = KC1 + KC2, KT1 = KC1 / 32, KT2 = KC2 / 32, KT = KT1 + KT2;
  constexpr int BN = 64, TN = 4, STAGE = BN * 64, DIST = RESIDENT ? 10 : 3, NSLOT = RESIDENT ? 10 : DIST + 2, RING = NSLOT * STAGE;
  constexpr int LDC = BN + 8, CW = 16 * LDC * 2, CPR = 8, RPI = 8, NIT = 2;
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + RING + 4 * CW);   // [4 waves][2][64]
  float* xcoef = red + 8 * BN;                                     // s2 | t2 | mean2 | dg | cn (the epilogue reads its coefficients from here)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int mb = xcd_remap(blockIdx.x, a.nMB);
  const T* __restrict__ gp = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ y2p = reinterpret_cast<const T*>(a.y2);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  T* __restrict__ dx = reinterpret_cast<T*>(a.dx);
  const bool full = (long long)(mb + 1) * BM <= a.M;
  const int li = lane & 15, gl = lane >> 4;

  for (int i = tid; i < 5 * KC2; i += 256) {
    const int which = i / KC2, c = i - which * KC2;
    xcoef[i] = which == 0 ? a.s2[c] : which == 1 ? a.t2[c] : which == 2 ? a.mean2[c] : which == 3 ? (a.dg ? a.dg[c] : 0.f) : a.cn[c];
  }

  // ---- the A operand: rows mb*128 + wid*32 + i*16 + li; chunks 0..7 from g, 8..9 from y2 (normalised below) ----
  const long long arow0 = (long long)mb * BM + widu * (16 * TM);
  uint4 areg[TM][KT];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    long long m = arow0 + i * 16 + li;
    if (m >= a.M) m = a.M - 1;
    const T* sg = gp + m * KC1 + gl * 8;
    const T* sy = y2p + m * KC2 + gl * 8;
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) areg[i][kt] = ld16_nt(sg + kt * 32);
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) areg[i][KT1 + kt] = *reinterpret_cast<const uint4*>(sy + kt * 32);
  }
  // ---- the epilogue's operands in its row-store layout: y2 (mask, sums) and, ACC, what dx holds ----
  const int er = lane / CPR, ec = lane % CPR;
  uint4 yv[TM * NIT], pv[ACC ? TM * NIT : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      long long m = arow0 + i * 16 + er + it * RPI;
      if (m >= a.M) m = a.M - 1;
      yv[i * NIT + it] = *reinterpret_cast<const uint4*>(y2p + m * KC2 + ec * 8);
      if constexpr (ACC) pv[i * NIT + it] = *reinterpret_cast<const uint4*>(dx + m * BN + ec * 8);
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();   // the coefficient table is in LDS (no LDS-DMA in flight yet)
#pragma unroll
  for (int kt = 0; kt < KT2; ++kt) {
    float qs[8], qt[8];
    const float* cs = xcoef + kt * 32 + gl * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qs[e] = cs[e];
      qt[e] = cs[KC2 + e];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      Vec16<T> v, w2;
      v.raw = areg[i][KT1 + kt];
      XfMath<T>::template run<false>(v, w2, qs, qt, nullptr, nullptr, 1, false);   // a2 = relu(r(y2*s2 + t2)): maai_bn_act_fwd
      areg[i][KT1 + kt] = v.raw;
    }
  }
  if (!full) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if (arow0 + i * 16 + li >= a.M) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) areg[i][kt] = make_uint4(0, 0, 0, 0);
      }
  }

  // ---- weight stages: K-step s = 32 input channels x 64 output channels, 64-byte rows, swizzled like conv_igemm ----
  const int r0 = tid >> 2;
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);
  const T* wsrc = w + (long long)r0 * K + chunk * 8;
  int islot = 0, issued = 0;
  auto issue_b = [&]() {
    dma16<true>(wsrc, smem + islot * STAGE + widu * 1024);
    ++issued;
    if (++islot == NSLOT) islot = 0;
    wsrc += 32;
  };
#pragma unroll
  for (int s = 0; s < DIST; ++s) issue_b();

  const int foff = li * 64 + ((gl ^ (((li >> 3) & 1) << 1)) << 4);
  char* cw = smem + RING + widu * CW;
  const uint32_t cwa = (uint32_t)(uintptr_t)(cw + ((gl * 4) * LDC + li) * 2);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int slot = 0;
  if constexpr (RESIDENT) {
    dfold_wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every stage has landed
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if constexpr (!RESIDENT) {
      const int rem = KT - 1 - kt;
      if (rem >= DIST - 1) dfold_wait_vm<DIST - 1>();
      else if (rem == 1) dfold_wait_vm<1>();
      else dfold_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (issued < KT) issue_b();   // into the slot read two steps ago
    }
    const char* sb = smem + slot * STAGE + foff;
    frag_t bfr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(sb + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const frag_t af = __builtin_bit_cast(frag_t, areg[i][kt]);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af, bfr[j], acc[i][j]);
    }
    if (++slot == NSLOT) slot = 0;
  }

  // ---- epilogue: wave-private C area -> rows of 8 channels per lane; + cn (+ previous), mask, sums, store ----
  // (the per-channel coefficients come from the LDS table only now: held in registers through the K loop they cost the 40
  //  registers that separate two workgroups per CU from three)
  float cnc[4], m2[8], sc2[8], sh2[8], dgr[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) cnc[j] = xcoef[4 * KC2 + j * 16 + (lane & 15)];   // C layout: this lane's column of accumulator tile j
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc2[e] = xcoef[ec * 8 + e];
    sh2[e] = xcoef[KC2 + ec * 8 + e];
    m2[e] = xcoef[2 * KC2 + ec * 8 + e];
    dgr[e] = xcoef[3 * KC2 + ec * 8 + e];
    s1[e] = 0.f;
    s2[e] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      // (the constant is added in fp32 BEFORE the rounding to bf16: the accumulator leaves the registers as acc + cn)
      const f32x4 v = acc[i][j] + (f32x4){cnc[j], cnc[j], cnc[j], cnc[j]};
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
    for (int it = 0; it < NIT; ++it) {
      const long long m = arow0 + i * 16 + er + it * RPI;
      const bool ok = full || m < a.M;
      Vec16<T> v, vy;
      v.load(reinterpret_cast<const T*>(cw) + (er + it * RPI) * LDC + ec * 8);
      vy.raw = yv[i * NIT + it];
      float fv[8], fy[8];
      v.get(fv);
      vy.get(fy);
#pragma unroll
      for (int e = 0; e < 8; ++e) fv[e] += dgr[e] * fmaxf(round_as<T>(fy[e] * sc2[e] + sh2[e]), 0.f);   // T's diagonal, in fp32
      if constexpr (ACC) {
        Vec16<T> vp;
        vp.raw = pv[i * NIT + it];
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
        v.store(dx + m * BN + ec * 8);
      }
    }
  }
  // ---- the unit below's sums: lanes l, l + 8, ... hold the same channels ----
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
      red[(widu * 2 + 0) * BN + lane * 8 + e] = s1[e];
      red[(widu * 2 + 1) * BN + lane * 8 + e] = s2[e];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < 2 * BN) {
    const int which = tid / BN, c = tid - which * BN;
    a.slab[((long long)mb * 2 + which) * BN + c] =
        red[which * BN + c] + red[(2 + which) * BN + c] + red[(4 + which) * BN + c] + red[(6 + which) * BN + c];
  }
}

extern "C" long long maai_conv_dfold_rows(long long M) { return (M + 127) / 128; }

// g [M][256], y2 [M][64] bf16; w = Wcat [64][320] bf16; cn, mean2, s2, t2 [64] fp32; dx [M][64] bf16 (accumulate != 0: += in
// place, before the mask; the sums are those of the stored result); slab [maai_conv_dfold_rows(M)][2][64] fp32.
extern "C" int maai_conv_dfold(const void* g, const void* y2, const void* w, const float* cn, const float* dg, const float* mean2,
                               const float* s2, const float* t2, void* dx, float* slab, long long M, int accumulate, void* stream) {
  MAAI_CHECK_ARG(g && y2 && w && cn && mean2 && s2 && t2 && dx && slab && M > 0, "conv_dfold: null pointer");
  MAAI_CHECK_ARG(M < (1ll << 31), "conv_dfold: pixel count must fit 31 bits");
  DfoldArgs a;
  a.g = g; a.y2 = y2; a.w = w; a.cn = cn; a.dg = dg; a.mean2 = mean2; a.s2 = s2; a.t2 = t2; a.dx = dx; a.slab = slab; a.M = M;
  a.nMB = (int)((M + 127) / 128);
  static const bool resident = !(getenv("MAAI_DFOLD_RESIDENT") && atoi(getenv("MAAI_DFOLD_RESIDENT")) == 0);   // A/B knob
  const int lds = (resident ? 10 : 5) * 64 * 64 + 4 * 16 * 72 * 2 + 8 * 64 * 4 + 5 * 64 * 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static int attr[4][64] = {{0}};
#define MAAI_DFOLD_LAUNCH(ACCV, RESV, SLOT)                                                                          \
  do {                                                                                                               \
    maai_ensure_lds(reinterpret_cast<const void*>(&conv_dfold_kernel<ACCV, RESV>), lds, attr[SLOT]);                 \
    MAAI_NOTE_KERNEL(conv_dfold_kernel<ACCV, RESV>);                                                                 \
    hipLaunchKernelGGL((conv_dfold_kernel<ACCV, RESV>), dim3((unsigned)a.nMB), dim3(256), lds, st, a);               \
  } while (0)
  if (accumulate) {
    if (resident) MAAI_DFOLD_LAUNCH(true, true, 0); else MAAI_DFOLD_LAUNCH(true, false, 1);
  } else {
    if (resident) MAAI_DFOLD_LAUNCH(false, true, 2); else MAAI_DFOLD_LAUNCH(false, false, 3);
  }
#undef MAAI_DFOLD_LAUNCH
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
