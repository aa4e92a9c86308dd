// The implicit-GEMM convolution kernel template and its launcher, shared by conv_fwd.hip (plain / epilogue-fused
// launches) and conv_xf.hip (launches whose A operand is normalised on the way in).  See conv_fwd.hip for the design.
#pragma once
#ifndef MAAI_EXP
#define MAAI_EXP 0   // experiment bits for A/B builds (scripts/build_variant.sh); 0 = the shipped kernel
#endif
#include "common.h"
#include "maai_internal.h"
#include <stdlib.h>

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  typedef f32x4 frag;
  // lane group g = lane>>4 holds k = 4g+j in element j of its 16-byte chunk; MFMA j
  // consumes element j of both operands, i.e. a consistent permutation of K.
  __device__ static __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};

static __device__ uint4 g_zero64[4];  // (one copy per translation unit) zero page for padding / out-of-range lanes of the LDS-DMA loads

// One 16-byte-per-lane LDS-DMA instruction (1 KiB per wave, lane-linear destination).  ASM: issued as inline assembly,
// invisible to hipcc — which otherwise orders EVERY LDS access it can see behind all LDS-DMA in flight with an
// s_waitcnt vmcnt(0); the kernels that rewrite staged operands in LDS (XF) could then keep nothing in flight.
// M0 (the DMA's LDS base) is compiler-owned: saved and restored inside the statement (cdna_hip_programming.md §5.7).
template <bool ASM>
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst) {
  if constexpr (ASM) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  } else {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
  }
}

struct ConvArgs {
  const void* x;
  const void* w;
  void* y;
  float* stats;
  const void* mask;  // optional: output *= (mask > 0), same layout as y (ReLU gradient of the tensor y is the gradient of)
  long long M;
  int N, IH, IW, Cin;
  int Cout, KH, KW;
  int stride, pad_h, pad_w;
  int OHg, OWg;
  int OH, OW;
  int ostr, ooh, oow;
  int accumulate;
  int nMB, nNB;
  // fused epilogues (maai_conv_epilogue): 0 store, 1 statistics only, 2 BN-apply(+residual)(+ReLU),
  // 3 BN-backward reduce (partials of dz and dz*(y-mean)), 4 BN-backward apply (k1*dz - k2 - k3*y)
  int emode, erelu;
  const float* ep0;
  const float* ep1;
  const float* ep2;
  const void* et;
  int mask_bits;  // mask is a 1-bit-per-element array (maai_bn_act_fwd_mask), EMODE 6 / 16-bit types only
  int sum_incr;   // EMODE 6 with accumulate: reduce the sums of (stored - previous content) instead of the stored value
  int tilesX, tilesY;  // HALO kernels: 16-wide x BM/16-high output patches per image
  // AXF kernels: the A operand is k1*x - k2 - k3*a2 per input channel (BatchNorm-backward apply of the layer above),
  // computed while staging; a_out (nullable) receives it for the weight gradient
  const void* a2;
  const float* ak1;
  const float* ak2;
  const float* ak3;
  void* a_out;
  // XF kernels: the A operand is act(x*xs[ci] + xt[ci]) — the BatchNorm apply (+ReLU) of the unit that produced x —
  // formed in LDS after the DMA lands, so the normalised activation never exists in HBM.  XF == 2: two raw tensors
  // meet, act((x*xs + xt) + r(xb*xs2 + xt2)) (the residual join of a bottleneck, r = rounding to the storage type);
  // x_out / x_bits (nullable) receive the joined activation and its 1-bit ReLU mask once (column tile 0).
  const float* xs;
  const float* xt;
  int x_relu;
  const void* xb;
  const float* xs2;
  const float* xt2;
  void* x_out;
  unsigned char* x_bits;
  // chained launches (conv_chain.hip): x is not read — it is recomputed as conv(pre_x [M][pre_cin], pre_w [Cin][pre_cin]),
  // pre_x normalised on load by pre_xs / pre_xt (nullable) — and joined with xb as above; pre_y_out (nullable) keeps it
  const void* pre_x;
  const void* pre_w;
  const float* pre_xs;
  const float* pre_xt;
  int pre_relu, pre_cin;
  void* pre_y_out;
  // two-source A operand (pointwise EMODE 6 launches: the folded unit's data gradient, engine._FOLD): input channels
  // [0, cin1) come from x (row pitch cin1), [cin1, Cin) from x2 (row pitch Cin - cin1); ebias (nullable, [Cout]) is added to
  // the accumulators before they are rounded
  const void* x2;
  int cin1;
  const float* ebias;
  const float* ediag;   // EMODE 6 (with ep1/ep2/et): out += ediag[c] * relu(r(et*ep1 + ep2)) — the folded data gradient's fp32 diagonal
};

// EMODE: 0 plain store, 1 statistics only, 2..4 fused BN epilogues, 5 store with accumulate and/or ReLU mask,
// 6 = 5 plus the BN-backward partial sums of the stored gradient (MAAI_EPI_DGRAD_REDUCE).
// PW: pointwise stride-1 layer (input pixel == output pixel): no row decode, no tap loop, no bounds tests.
// HALO: 3x3 stride-1 same-size layers.  The M tile is a 16-wide x BM/16-high patch of output pixels of ONE image
// and the A operand is its (BM/16+2) x 18 input halo, staged once per 32-channel chunk and read by all nine taps
// at shifted pixel addresses, instead of nine separately staged 64-byte row sets: 2.3-3.4x fewer LDS-DMA bytes per
// MFMA on the layers whose K loop is bound by exactly that traffic.  K order: chunk-major, tap-minor.
// AXF (pointwise data gradients): A = k1*dz - k2 - k3*y — the BatchNorm-backward apply of the layer whose gradient
// this convolution propagates — is computed on the way into LDS (global -> registers -> LDS, two slots) instead of
// being written by a separate pass and read back; the transformed operand is also stored once (column tile 0) for
// the weight gradient.  Same arithmetic, same bits as maai_bn_act_bwd_apply followed by the plain kernel.
template <typename T, int BM, int BN, int NSTAGE, int EMODE, bool PW, bool HALO = false, bool AXF = false, int XF = 0>
__global__ __launch_bounds__(256, (BM == 256 || BN == 256) ? 2 : 3) void conv_igemm_kernel(ConvArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 4 * EPC;               // 64-byte rows
  constexpr int WGM = (BM == 256 && BN == 64) ? 4 : 2, WGN = 4 / WGM;  // wave grid: 2x2, or 4x1 for the 256x64 tile
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int AR = BM / 64, BR = BN / 64;  // rows staged per thread
  constexpr int STAGE = HALO ? BN * 64 : (BM + BN) * 64;  // bytes per ring buffer (HALO: weights only)
  constexpr int TH = BM / 16;                // HALO: patch height; halo image = (TH+2) rows x 24 pixel slots (18 used)
  constexpr int HROWS = (TH + 2) * 24;       //       pixel slots of 64 bytes
  constexpr int NH = (HROWS + 63) / 64;      //       LDS-DMA instructions per thread per halo
  constexpr int HSLOT = NH * 4096;           //       bytes per halo buffer (two of them after the weight ring)
  constexpr int LDC = BN + EPC;              // C-tile row pitch (elements), 16-B padded
  constexpr int RING = HALO ? NSTAGE * STAGE + 2 * HSLOT : NSTAGE * STAGE;  // XF: the second-operand ring and the coefficient table follow
  constexpr int XBS = BM * 64;               // XF == 2: bytes per stage of the second operand's ring
  static_assert(XF == 0 || (!AXF && (EMODE == 0 || EMODE == 2)), "XF: forward convolutions");
  static_assert(XF != 2 || (PW && !HALO), "XF == 2: pointwise layers");
  typedef typename Mma<T>::frag frag_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int logical = xcd_remap(blockIdx.x, a.nMB * a.nNB);
  const int mb = logical / a.nNB, nb = logical - mb * a.nNB;
  // HALO: patch origin of this tile
  int hn = 0, oy0 = 0, ox0 = 0;
  if constexpr (HALO) {
    const int tpi = a.tilesX * a.tilesY;
    hn = mb / tpi;
    const int rem = mb - hn * tpi;
    const int tyi = rem / a.tilesX;
    oy0 = tyi * TH;
    ox0 = (rem - tyi * a.tilesX) * 16;
  }
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  const int K = a.KH * a.KW * a.Cin;
  const int r0 = tid >> 2;
  // LDS slot (row r, chunk tid&3) holds source chunk (tid&3) ^ swz(r): inverse swizzle on the SOURCE
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);

  // ---- per-thread row decode (fixed for the whole K loop); 32-bit arithmetic (M < 2^31 is checked on the host),
  //      and no decode at all for pointwise stride-1 layers where the input pixel IS the output pixel ----
  long long abase[AR];
  long long abase2[AR];   // two-source operand (pointwise EMODE 6): the row's offset in x2, minus cin1 (so that + kt*BK indexes it)
  int ihb[AR], iwb[AR];
  const unsigned ohw = (unsigned)(a.OHg * a.OWg);
  constexpr bool pointwise = PW;
#pragma unroll
  for (int i = 0; i < (HALO ? 0 : AR); ++i) {
    const long long m = (long long)mb * BM + r0 + 64 * i;
    if (m < a.M) {
      if (pointwise) {
        ihb[i] = 0;
        iwb[i] = 0;
        abase[i] = m * (a.x2 ? a.cin1 : a.Cin) + chunk * EPC;
        if constexpr (PW && EMODE == 6 && XF == 0) abase2[i] = m * (a.Cin - a.cin1) + chunk * EPC - a.cin1;
      } else {
        const unsigned mu = (unsigned)m;
        const unsigned n = mu / ohw;
        const unsigned rem = mu - n * ohw;
        const unsigned oh = rem / (unsigned)a.OWg, ow = rem - oh * (unsigned)a.OWg;
        ihb[i] = (int)oh * a.stride - a.pad_h;
        iwb[i] = (int)ow * a.stride - a.pad_w;
        abase[i] = (((long long)n * a.IH + ihb[i]) * a.IW + iwb[i]) * a.Cin + chunk * EPC;
      }
    } else {
      ihb[i] = -(1 << 28);
      iwb[i] = -(1 << 28);
      abase[i] = 0;
    }
  }
  const T* wp[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) wp[i] = w + (long long)(nb * BN + r0 + 64 * i) * K + chunk * EPC;

  int kh = 0, kw = 0, c0 = 0;  // position of the NEXT stage to issue
  const int KT = K / BK;
  constexpr int NL = HALO ? BR : (XF == 2 ? 2 * AR + BR : AR + BR);  // LDS-DMA instructions per thread per stage (HALO: the
                                           // halo's own, once per nine stages, only make the counted waits below conservative)
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const T* zsrc = reinterpret_cast<const T*>(g_zero64);

  auto issue_stage = [&](int kt, int slot) {
    if constexpr (HALO) {
      // kh = chunk counter, kw = tap counter of the NEXT stage to issue
      if (kw == 0) {
        char* hb = smem + NSTAGE * STAGE + (kh & 1) * HSLOT + widu * 1024;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
          const int hp = r0 + 64 * i;                 // pixel slot of this lane
          const int hy = hp / 24, hx = hp - hy * 24;
          const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
          const bool ok = hx < 18 && hy < TH + 2 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
          const int sc = (tid & 3) ^ (((hp >> 2) & 1) << 1);   // slot chunk (tid&3) holds source chunk sc
          const T* src = ok ? x + (((long long)hn * a.IH + iy) * a.IW + ix) * a.Cin + kh * BK + sc * EPC : zsrc;
          dma16<XF != 0>(src, hb + i * 4096);
        }
      }
      char* sb = smem + slot * STAGE + widu * 1024;
      const long long boff = (long long)kw * a.Cin + kh * BK;
#pragma unroll
      for (int i = 0; i < BR; ++i)
        dma16<XF != 0>((wp[i] + boff), sb + i * 4096);
      if (++kw == 9) { kw = 0; ++kh; }
      return;
    }
    const long long tapoff = PW ? (long long)kt * BK : ((long long)kh * a.IW + kw) * a.Cin + c0;
    char* sa = smem + slot * STAGE + widu * 1024;
    char* sb = sa + BM * 64;
    // (two-source operand, pointwise EMODE 6 only: K-steps past cin1 read the second tensor, whose rows are Cin - cin1 long)
    const bool second = PW && EMODE == 6 && XF == 0 && a.x2 != nullptr && kt * BK >= a.cin1;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = PW ? (ihb[i] >= 0) : ((unsigned)(ihb[i] + kh) < (unsigned)a.IH && (unsigned)(iwb[i] + kw) < (unsigned)a.IW);
      const T* src = ok ? (x + abase[i] + tapoff) : zsrc;
      if (second && ok) src = reinterpret_cast<const T*>(a.x2) + abase2[i] + (long long)kt * BK;
      dma16<XF != 0>(src, sa + i * 4096);
    }
    if constexpr (XF == 2) {
      const T* __restrict__ xb = reinterpret_cast<const T*>(a.xb);
      char* s2 = smem + RING + slot * XBS + widu * 1024;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const T* src = (ihb[i] >= 0) ? (xb + abase[i] + tapoff) : zsrc;
        dma16<XF != 0>(src, s2 + i * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i)
      dma16<XF != 0>((wp[i] + (long long)kt * BK), sb + i * 4096);
    if constexpr (!PW) {
      c0 += BK;
      if (c0 >= a.Cin) {
        c0 = 0;
        if (++kw >= a.KW) { kw = 0; ++kh; }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offset inside a 16-row group (swizzled): row = lane&15, chunk = lane>>4
  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) << 1)) << 4);

  // HALO: per-lane fragment offsets inside a halo row for kw = 0, 1, 2 (24 slots per row keep bit 2 of the pixel
  // slot, which the swizzle uses, a function of tx + kw alone); ctap / cchunk = tap and chunk of the stage being
  // multiplied
  int hoff[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int px = frow + q;
    hoff[q] = px * 64 + (((lane >> 4) ^ (((px >> 2) & 1) << 1)) << 4);
  }
  int ctap = 0, cchunk = 0;
  if constexpr (AXF) {
    static_assert(!AXF || (PW && !HALO && sizeof(T) == 2 && BM == 128), "AXF: pointwise bf16 128-row tiles");
    constexpr int SLOT = (BM + BN) * 64;
    float* coef = reinterpret_cast<float*>(smem + 2 * SLOT);  // k1 | k2 | k3, K floats each
    for (int i = tid; i < 3 * K; i += 256) coef[i] = i < K ? a.ak1[i] : (i < 2 * K ? a.ak2[i - K] : a.ak3[i - 2 * K]);
    const T* __restrict__ y2 = reinterpret_cast<const T*>(a.a2);
    T* __restrict__ dyo = reinterpret_cast<T*>(a.a_out);
    const bool keep_dy = dyo != nullptr && nb == 0;
    Vec16<T> rz[AR], ry[AR];
    auto load_a = [&](int kt) {
#pragma unroll
      for (int i = 0; i < AR; ++i)
        if (ihb[i] >= 0) {
          rz[i].load(x + abase[i] + (long long)kt * BK);
          ry[i].load(y2 + abase[i] + (long long)kt * BK);
        }
    };
    auto issue_b = [&](int kt, int slot) {
      char* sb = smem + slot * SLOT + BM * 64 + widu * 1024;
#pragma unroll
      for (int i = 0; i < BR; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp[i] + (long long)kt * BK),
                                         (__attribute__((address_space(3))) void*)(sb + i * 4096), 16, 0, 0);
    };
    auto store_a = [&](int kt, int slot) {
      const float* c1 = coef + kt * BK + chunk * EPC;
      float q1[8], q2[8], q3[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q1[e] = c1[e];
        q2[e] = c1[K + e];
        q3[e] = c1[2 * K + e];
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        float d[8], yy[8];
        Vec16<T> v;
        if (ihb[i] >= 0) {
          rz[i].get(d);
          ry[i].get(yy);
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = q1[e] * d[e] - q2[e] - q3[e] * yy[e];
          v.set(d);
          if (keep_dy) v.store(dyo + abase[i] + (long long)kt * BK);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = 0.f;
          v.set(d);
        }
        v.store(reinterpret_cast<T*>(smem + slot * SLOT + (r0 + 64 * i) * 64 + (tid & 3) * 16));
      }
    };
    load_a(0);
    issue_b(0, 0);
    __syncthreads();  // coefficients are in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_a(0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int slot = kt & 1;
      if (kt + 1 < KT) {  // slot^1 was last read in iteration kt-1, whose closing barrier everyone has passed
        load_a(kt + 1);
        issue_b(kt + 1, slot ^ 1);
      }
      const char* sa = smem + slot * SLOT + (wm * WM) * 64 + foff;
      const char* sb = smem + slot * SLOT + BM * 64 + (wn * WN) * 64 + foff;
      frag_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag_t*>(sa + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(sb + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af[i], bfr[j], acc[i][j]);
      if (kt + 1 < KT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_a(kt + 1, slot ^ 1);
      }
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  // ---- XF: the staged A operand is normalised (+ activated) IN LDS by the thread whose DMA brought it ----
  // Each thread owns the 16-byte chunks its own LDS-DMA instructions wrote (lane-linear destination), so once its
  // own counted vmcnt has retired a stage it can rewrite them in place without any other synchronisation; the K
  // loop's next barrier publishes them.  Padding / out-of-range chunks (zero page) stay zero: the convolution pads
  // the ACTIVATION, not the raw tensor.  Same arithmetic, in the same order, as maai_bn_act_fwd / _fwd2.
  float* xcoef = reinterpret_cast<float*>(smem + RING + (XF == 2 ? NSTAGE * XBS : 0));  // xs | xt (| xs2 | xt2), Cin floats each
  int tkh = 0, tkw = 0, tc0 = 0;  // row-staged layers: tap and channel offset of the next stage to transform
  // one stage's coefficients live in registers while its chunks are rewritten (every chunk of a thread carries the
  // same channels)
  struct XfCoef {
    float s[EPC], t[EPC], s2[EPC], t2[EPC];
  };
  auto xf_load = [&](const float* cs, XfCoef& q) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      q.s[e] = cs[e];
      q.t[e] = cs[a.Cin + e];
    }
    if constexpr (XF == 2) {
      if (a.xs2) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          q.s2[e] = cs[2 * a.Cin + e];
          q.t2[e] = cs[3 * a.Cin + e];
        }
      }
    }
  };
  auto xf_chunk = [&](T* p, const T* pb, const XfCoef& q, long long goff, bool side) {
    Vec16<T> v, w2;
    v.load(p);
    if constexpr (XF == 2) w2.load(pb);
    const bool wb = XF == 2 && side && a.x_bits != nullptr;
    const unsigned b = XfMath<T>::template run<XF == 2>(v, w2, q.s, q.t, (XF == 2 && a.xs2) ? q.s2 : nullptr, q.t2, a.x_relu, wb);
    v.store(p);
    if constexpr (XF == 2) {
      if (side) {
        v.store_nt(reinterpret_cast<T*>(a.x_out) + goff);
        if (wb) a.x_bits[goff >> 3] = (unsigned char)b;
      }
    }
  };
  auto xform_rows = [&](int kt, int slot) {
    XfCoef q;
    xf_load(xcoef + (PW ? kt * BK : tc0) + chunk * EPC, q);
    const bool side = XF == 2 && a.x_out != nullptr && nb == 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = PW ? (ihb[i] >= 0) : ((unsigned)(ihb[i] + tkh) < (unsigned)a.IH && (unsigned)(iwb[i] + tkw) < (unsigned)a.IW);
      if (ok)
        xf_chunk(reinterpret_cast<T*>(smem + slot * STAGE + i * 4096 + tid * 16),
                 reinterpret_cast<const T*>(smem + RING + slot * XBS + i * 4096 + tid * 16), q,
                 PW ? abase[i] + (long long)kt * BK : 0, side);
    }
    if constexpr (!PW) {
      tc0 += BK;
      if (tc0 >= a.Cin) {
        tc0 = 0;
        if (++tkw >= a.KW) { tkw = 0; ++tkh; }
      }
    }
  };
  auto xform_halo = [&](int c) {  // the halo of 32-channel chunk c
    char* hb = smem + NSTAGE * STAGE + (c & 1) * HSLOT;
    const int sc = (tid & 3) ^ (((r0 >> 2) & 1) << 1);
    XfCoef q;
    xf_load(xcoef + c * BK + sc * EPC, q);
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int hp = r0 + 64 * i;
      const int hy = hp / 24, hx = hp - hy * 24;
      const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
      const bool ok = hx < 18 && hy < TH + 2 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      if (ok) xf_chunk(reinterpret_cast<T*>(hb + i * 4096 + tid * 16), nullptr, q, 0, false);
    }
  };
  if constexpr (XF != 0) {
    const int nco = (XF == 2 && a.xs2) ? 4 * a.Cin : 2 * a.Cin;
    for (int i = tid; i < nco; i += 256) {
      const int which = i / a.Cin, c = i - which * a.Cin;
      xcoef[i] = which == 0 ? a.xs[c] : (which == 1 ? a.xt[c] : (which == 2 ? a.xs2[c] : a.xt2[c]));
    }
    __syncthreads();  // before any LDS-DMA is in flight (a later __syncthreads would drain the ring)
  }
  const int pre = KT < NSTAGE - 1 ? KT : NSTAGE - 1;
  for (int s = 0; s < pre; ++s) issue_stage(s, s);
  if constexpr (XF != 0) {
    if (pre >= 3) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
    } else if (pre == 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (HALO) xform_halo(0); else xform_rows(0, 0);
  }

  for (int kt = 0; kt < KT; ++kt) {
    if constexpr (XF != 0) {
      // stage kt was retired and transformed at the end of the previous iteration: only the LDS writes are pending
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
    // stages kt+1 .. min(KT-1, kt+2) may stay in flight; stage kt must have landed
    const int ahead = KT - 1 - kt;
    if (NSTAGE >= 4 && ahead >= 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
    } else if (NSTAGE >= 3 && ahead >= 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
    __builtin_amdgcn_s_barrier();
    // everyone has finished reading slot (kt-1)%NSTAGE -> refill it with stage kt+NSTAGE-1
    if (kt + NSTAGE - 1 < KT) issue_stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const int slot = kt % NSTAGE;
    const char* sa;
    const char* sb;
    if constexpr (HALO) {
      const int ckh = ctap / 3, ckw = ctap - ckh * 3;
      const int ho = ckw == 0 ? hoff[0] : (ckw == 1 ? hoff[1] : hoff[2]);
      sa = smem + NSTAGE * STAGE + (cchunk & 1) * HSLOT + (wm * (WM / 16) + ckh) * (24 * 64) + ho;
      sb = smem + slot * STAGE + (wn * WN) * 64 + foff;
      if (++ctap == 9) { ctap = 0; ++cchunk; }
    } else {
      sa = smem + slot * STAGE + (wm * WM) * 64 + foff;
      sb = smem + slot * STAGE + BM * 64 + (wn * WN) * 64 + foff;
    }
    frag_t af[TM], bfr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag_t*>(sa + i * (HALO ? 24 * 64 : 16 * 64));
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(sb + j * 16 * 64);
#if MAAI_EXP & 1
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af[i], bfr[j], acc[i][j]);
#if MAAI_EXP & 1
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (XF != 0) {
      if (kt + 1 < KT) {
        // behind this step's MFMAs: retire stage kt+1 (stages kt+2 .. may stay in flight) and transform it
        const int ahead = KT - 2 - kt;
        if (NSTAGE >= 4 && ahead >= 2) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
        } else if (NSTAGE >= 3 && ahead >= 1) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (HALO) {
          if (ctap == 0) xform_halo(cchunk);  // (ctap, cchunk) already name stage kt+1
        } else {
          xform_rows(kt + 1, (kt + 1) % NSTAGE);
        }
      }
    }
  }
  }
  __syncthreads();  // all DMA retired (vmcnt(0) above); the ring is now reused as the C tile
  if constexpr (HALO) {
    // patch rows / columns outside the image hold sums over real neighbours: zero them so that neither the
    // statistics nor anything else sees them (their stores are skipped below)
    if (oy0 + TH > a.OH || ox0 + 16 > a.OW) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bool rowok = oy0 + wm * (WM / 16) + i < a.OH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rowok && (ox0 + (lane >> 4) * 4 + r < a.OW);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (!ok) acc[i][j][r] = 0.f;
        }
      }
    }
  }

  // ---- epilogue ----
  T* ct = reinterpret_cast<T*>(smem);
  constexpr int CROWS = BN > 128 ? 64 : 128;  // rows the C tile holds; taller (or 256-column) tiles drain in phases
  constexpr int NPH = BM / CROWS > 0 ? (BM + CROWS - 1) / CROWS : 1;
  float* red = reinterpret_cast<float*>(smem + (BM < CROWS ? BM : CROWS) * LDC * (int)sizeof(T));  // [WGM wm x 4 lane groups][2][BN]
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  constexpr int CPR = BN / EPC;  // 16-byte chunks per tile row
  constexpr int NV = Vec16<T>::N;
  const bool dense = (a.ostr == 1 && a.ooh == 0 && a.oow == 0 && a.OHg == a.OH && a.OWg == a.OW);
  const int chf = tid % CPR;                 // this thread's chunk column (256 % CPR == 0)
  const int cch0 = nb * BN + chf * EPC;      // its first output channel
  constexpr bool FUSED = EMODE >= 2 && EMODE <= 4;  // (EMODE 5 = accumulate / mask store, not a BN epilogue)
  constexpr bool PARAMS = FUSED || EMODE == 6;
  constexpr int NQ = PARAMS ? NV : 1;        // per-channel epilogue parameters live only in the fused kernels
  float q0[NQ], q1[NQ], q2[NQ], s1[NQ], s2[NQ];
  if constexpr (PARAMS) {
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      q0[e] = a.ep0 ? a.ep0[cch0 + e] : (EMODE == 2 ? 1.f : 0.f);
      q1[e] = ((EMODE == 2 || EMODE == 4 || EMODE == 6) && a.ep1) ? a.ep1[cch0 + e] : 0.f;
      q2[e] = ((EMODE == 4 || EMODE == 6) && a.ep2) ? a.ep2[cch0 + e] : 0.f;
      s1[e] = 0.f;
      s2[e] = 0.f;
    }
  }
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph) {
  if (ph > 0) __syncthreads();  // the previous phase's readers are done with the C tile
  if constexpr (EMODE == 1) {
    // statistics only: the accumulators never leave the registers
  } else if constexpr (sizeof(T) == 2) {
    // The short-K layers are VALU-bound in this epilogue (measured: ~600 vector instructions per wave against
    // 32 MFMAs), so the C tile is written with the fewest vector instructions: one v_cvt_pk_bf16_f32 per row
    // pair and four 16-bit LDS stores (low half / d16_hi) per 16x16 tile — no lane exchange, no selects, and
    // every address is one per-lane base plus a compile-time offset.
    unsigned short* cbase = reinterpret_cast<unsigned short*>(smem) + ((wm * WM) % CROWS + (lane >> 4) * 4) * LDC + wn * WN + (lane & 15);
    float cbias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cbias[j] = (EMODE == 6 && a.ebias) ? a.ebias[nb * BN + wn * WN + j * 16 + (lane & 15)] : 0.f;
    if (NPH == 1 || (wm * WM) / CROWS == ph) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4 v = acc[i][j];
          if constexpr (EMODE == 6) v += (f32x4){cbias[j], cbias[j], cbias[j], cbias[j]};
          const uint32_t p01 = pack_bf16x2(v[0], v[1]);
          const uint32_t p23 = pack_bf16x2(v[2], v[3]);
          // one v_cvt_pk per row PAIR: the low half goes out with ds_write_b16, the high half with its d16_hi form
          // (left to the compiler this became one conversion per value)
          const uint32_t ca = (uint32_t)(uintptr_t)cbase;   // one base register; the tile position is an immediate offset
          asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(ca), "v"(p01),
                       "n"((i * 16 * LDC + j * 16) * 2), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 2)
                       : "memory");
          asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(ca), "v"(p23),
                       "n"((i * 16 * LDC + j * 16) * 2 + LDC * 4), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 6)
                       : "memory");
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ml = (wm * WM) % CROWS + i * 16 + (lane >> 4) * 4 + r;
          if (NPH > 1 && (wm * WM) / CROWS != ph) continue;
          const int nl = wn * WN + j * 16 + (lane & 15);
          Store<T>::st(ct + ml * LDC + nl, acc[i][j][r]);
        }
  }
  const bool fstats = a.stats && EMODE != 3 && EMODE != 6 && ph == 0;
  if (fstats) {
    // per lane: column n = j*16 + (lane&15), rows of its lane group; the 8 (wm, lane-group) partials per column
    // meet in LDS (cheaper than 16 cross-row shuffles per wave)
    float* sred = red + ((wm * 4 + (lane >> 4)) * 2) * BN + wn * WN + (lane & 15);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): two rows per instruction, half the vector-issue slots
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i) {
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
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (fstats) {
#pragma unroll
    for (int o = tid; o < 2 * BN; o += 256) {
      const int which = o / BN, c = o - which * BN;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 4 * WGM; ++k) t += red[(k * 2 + which) * BN + c];
      a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] = t;
    }
  }
  if constexpr (EMODE == 1) return;
  constexpr int RPI = 256 / CPR;             // tile rows covered per iteration
  const bool full = dense && ((long long)(mb + 1) * BM <= a.M);
  const long long off0 = ((long long)mb * BM + ph * CROWS + tid / CPR) * a.Cout + nb * BN + chf * EPC;
  const long long ostep = (long long)RPI * a.Cout;
  if constexpr (EMODE == 5 || EMODE == 6) {
    // Read-modify-write epilogues.  The stores of one iteration may alias the loads of the next as far as the
    // compiler can tell, which would serialise eight load -> store round trips per thread; so the global loads
    // of NB iterations (previous content, the lower layer's y, the mask) are issued together, then consumed.
    constexpr int NIT = (BM < CROWS ? BM : CROWS) / RPI;
    constexpr int NB = (NIT % 4 == 0 && BM < 256) ? 4 : (NIT % 2 == 0 ? 2 : 1);  // (4 would spill under the 256-row tile)
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += NB) {
      long long ooffs[NB];
      bool ok[NB];
      Vec16<T> vo[NB], vy[NB], vm[NB];
      unsigned mb8[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int it = it0 + b;
        const int row = tid / CPR + it * RPI;
        long long ooff = off0 + it * ostep;
        ok[b] = true;
        if constexpr (HALO) {
          const int R = ph * CROWS + row;
          const int oy = oy0 + (R >> 4), ox = ox0 + (R & 15);
          ok[b] = oy < a.OH && ox < a.OW;
          ooff = (((long long)hn * a.OH + oy) * a.OW + ox) * a.Cout + nb * BN + chf * EPC;
        } else if (!full) {
          const long long m = (long long)mb * BM + ph * CROWS + row;
          ok[b] = m < a.M;
          long long opix = m;
          if (!dense) {
            const unsigned mu = (unsigned)m;
            const unsigned n = mu / ohw;
            const unsigned rem = mu - n * ohw;
            const unsigned oh = rem / (unsigned)a.OWg, ow = rem - oh * (unsigned)a.OWg;
            opix = ((long long)n * a.OH + oh * a.ostr + a.ooh) * a.OW + ow * a.ostr + a.oow;
          }
          ooff = opix * a.Cout + nb * BN + chf * EPC;
        }
        ooffs[b] = ooff;
        mb8[b] = 0;
        if (ok[b]) {
          if (a.accumulate) vo[b].load(y + ooff);
          if constexpr (EMODE == 6) {
            // (et null: the unit below needs the sum of the stored gradient only — its BatchNorm backward is folded through
            //  its convolution and never reads its raw output; the second sum is then meaningless and ignored)
            if (a.et) vy[b].load(reinterpret_cast<const T*>(a.et) + ooff); else vy[b].zero();
          }
          if (a.mask) {
            if (EMODE == 6 && NV == 8 && a.mask_bits)
              mb8[b] = reinterpret_cast<const unsigned char*>(a.mask)[ooff >> 3];
            else
              vm[b].load(reinterpret_cast<const T*>(a.mask) + ooff);
          }
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!ok[b]) continue;
        const int row = tid / CPR + (it0 + b) * RPI;
        Vec16<T> v;
        v.load(ct + row * LDC + chf * EPC);
        float fv[NV], fo[NV];
        v.get(fv);
        if (a.accumulate) {
          vo[b].get(fo);
#pragma unroll
          for (int e = 0; e < NV; ++e) fv[e] += fo[e];
        }
        float fy[NV];
        if constexpr (EMODE == 6) {
          vy[b].get(fy);
          if (a.ediag) {   // (before the previous content and the mask: it is part of this launch's product)
#pragma unroll
            for (int e = 0; e < NV; ++e) {
              const float act = fmaxf(round_as<T>(fy[e] * q1[e] + q2[e]), 0.f);
              fv[e] += a.ediag[cch0 + e] * act;
            }
          }
        }
        if (a.mask) {
          if (EMODE == 6 && NV == 8 && a.mask_bits) {
#pragma unroll
            for (int e = 0; e < NV; ++e) fv[e] = ((mb8[b] >> e) & 1u) ? fv[e] : 0.f;
          } else {
            float fm[NV];
            vm[b].get(fm);
#pragma unroll
            for (int e = 0; e < NV; ++e) fv[e] = fm[e] > 0.f ? fv[e] : 0.f;
          }
        } else if (EMODE == 6 && a.ep1 && a.ep2) {  // the lower layer's ReLU output is positive exactly where y*scale + shift is
#pragma unroll
          for (int e = 0; e < NV; ++e) fv[e] = (fy[e] * q1[e] + q2[e]) > 0.f ? fv[e] : 0.f;
        }
        v.set(fv);
        if constexpr (EMODE == 6) {
          v.get(fv);  // the rounded value being stored is what a separate reduction pass would read back
          if (a.accumulate && a.sum_incr) {
            // sums of the CHANGE this launch makes to the tensor (a strided second pass over a tensor whose first pass
            // already reduced its own values): the two slabs add up to the sums of the final tensor exactly
#pragma unroll
            for (int e = 0; e < NV; ++e) fv[e] -= fo[e];
          }
#pragma unroll
          for (int e = 0; e < NV; ++e) {
            s1[e] += fv[e];
            s2[e] += fv[e] * (fy[e] - q0[e]);
          }
        }
        v.store(y + ooffs[b]);
      }
    }
  } else {
#pragma unroll
  for (int it = 0; it < (BM < CROWS ? BM : CROWS) / RPI; ++it) {
    const int row = tid / CPR + it * RPI, ch = chf;
    long long opix;
    long long ooff_fast = off0 + it * ostep;
    if constexpr (HALO) {
      const int R = ph * CROWS + row;
      const int oy = oy0 + (R >> 4), ox = ox0 + (R & 15);
      if (oy >= a.OH || ox >= a.OW) continue;
      ooff_fast = (((long long)hn * a.OH + oy) * a.OW + ox) * a.Cout + nb * BN + ch * EPC;
    } else if (!full) {
      const long long m = (long long)mb * BM + ph * CROWS + row;
      if (m >= a.M) continue;
      opix = m;
      if (!dense) {
        const unsigned mu = (unsigned)m;
        const unsigned n = mu / ohw;
        const unsigned rem = mu - n * ohw;
        const unsigned oh = rem / (unsigned)a.OWg, ow = rem - oh * (unsigned)a.OWg;
        opix = ((long long)n * a.OH + oh * a.ostr + a.ooh) * a.OW + ow * a.ostr + a.oow;
      }
      ooff_fast = opix * a.Cout + nb * BN + ch * EPC;
    }
    const long long ooff = ooff_fast;
    T* dst = y + ooff;
    Vec16<T> v;
    v.load(ct + row * LDC + ch * EPC);
    if constexpr (FUSED) {
      float fv[NV];
      v.get(fv);
      if constexpr (EMODE == 2) {  // out = act(y*scale + shift (+ residual))
#pragma unroll
        for (int e = 0; e < NV; ++e) fv[e] = fv[e] * q0[e] + q1[e];
        if (a.et) {
          Vec16<T> r;
          r.load(reinterpret_cast<const T*>(a.et) + ooff);
          float fr[NV];
          r.get(fr);
#pragma unroll
          for (int e = 0; e < NV; ++e) fv[e] += fr[e];
        }
        if (a.erelu) {
#pragma unroll
          for (int e = 0; e < NV; ++e) fv[e] = fmaxf(fv[e], 0.f);
        }
        v.set(fv);
        v.store(dst);
      } else {
        Vec16<T> dzv;
        dzv.load(reinterpret_cast<const T*>(a.et) + ooff);
        float dz[NV];
        dzv.get(dz);
        if constexpr (EMODE == 3) {  // sums of dz and dz*(y-mean)
#pragma unroll
          for (int e = 0; e < NV; ++e) {
            s1[e] += dz[e];
            s2[e] += dz[e] * (fv[e] - q0[e]);
          }
        } else {  // dy = k1*dz - k2 - k3*y
#pragma unroll
          for (int e = 0; e < NV; ++e) fv[e] = q0[e] * dz[e] - q1[e] - q2[e] * fv[e];
          v.set(fv);
          v.store(dst);
        }
      }
      continue;
    }
    v.store(dst);
  }
  }
  }  // phases
  if constexpr (EMODE == 3 || EMODE == 6) {
    // lanes l, l+CPR, l+2CPR.. of a wave hold the same channels: butterfly, then the four waves through LDS
#pragma unroll
    for (int e = 0; e < NV; ++e) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o);
        s2[e] += __shfl_xor(s2[e], o);
      }
    }
    __syncthreads();  // everyone is done reading the C tile / the forward-statistics scratch
    float* red4 = reinterpret_cast<float*>(smem);  // [4 waves][2][BN], reuses the C tile
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        red4[(wid * 2 + 0) * BN + lane * EPC + e] = s1[e];
        red4[(wid * 2 + 1) * BN + lane * EPC + e] = s2[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = tid; o < 2 * BN; o += 256) {
      const int which = o / BN, c = o - which * BN;
      a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] =
          red4[which * BN + c] + red4[(2 + which) * BN + c] + red4[(4 + which) * BN + c] + red4[(6 + which) * BN + c];
    }
  }
}

template <typename T, int BM, int BN, int NSTAGE, int EMODE, bool PW, bool HALO = false, bool AXF = false, int XF = 0>
static int launch_conv_p(const ConvArgs& a, hipStream_t st) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int stage = HALO ? NSTAGE * BN * 64 + 2 * (((BM / 16 + 2) * 24 + 63) / 64) * 4096 : NSTAGE * (BM + BN) * 64;
  constexpr int crows = BN > 128 ? 64 : 128;
  constexpr int epi = (BM < crows ? BM : crows) * (BN + EPC) * (int)sizeof(T) + 32 * BN * (int)sizeof(float);
  const int axf_lds = AXF ? 2 * (BM + BN) * 64 + 12 * a.KH * a.KW * a.Cin : 0;  // two slots + k1|k2|k3
  // XF: second-operand ring (XF == 2) and the coefficient table behind the ring
  const int xf_lds = XF ? stage + (XF == 2 ? NSTAGE * BM * 64 + 16 * a.Cin : 8 * a.Cin) : 0;
  const int lds0 = stage > epi ? stage : epi;
  const int lds = AXF ? (axf_lds > epi ? axf_lds : epi) : (xf_lds > lds0 ? xf_lds : lds0);
  if (lds > 160 * 1024) {
    maai_set_error("conv2d_igemm: the transformed-operand tables do not fit in LDS for this many input channels");
    return MAAI_ERR_UNSUPPORTED;
  }
  static int attr_lds[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, NSTAGE, EMODE, PW, HALO, AXF, XF>), lds, attr_lds);
  const long long grid = (long long)a.nMB * a.nNB;
  MAAI_NOTE_KERNEL(conv_igemm_kernel<T, BM, BN, NSTAGE, EMODE, PW, HALO, AXF, XF>);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, NSTAGE, EMODE, PW, HALO, AXF, XF>), dim3((unsigned)grid), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}


// tile selection handed from conv_fwd.hip (which owns the shape rules) to conv_xf.hip (which owns the instantiations
// of the normalise-on-load kernels)
struct ConvSel {
  int dtype, bm, bn, nstage;
  bool halo, pw;
};
int maai_conv_xf_launch(const ConvArgs& a, const ConvSel& sel, hipStream_t st);
// conv_pws.hip: the streaming kernel for pointwise layers with Cin <= 256 (A operand in registers, weights streamed)
int maai_conv_pws_launch(const ConvArgs& a, hipStream_t st);
// conv_chain.hip: conv3 (recomputed) + BatchNorm + residual join + the next block's conv1 in one launch
int maai_conv_chain_launch(const ConvArgs& a, hipStream_t st);
// conv_pp.hip: the 8-wave ping-pong kernel (256 x 256 tiles, 64-deep K-tiles) for the MFMA-bound layers
bool maai_conv_pp_supported(const ConvArgs& a, int dtype);
int maai_conv_pp_rows(int Cout);
// conv_c64.hip: persistent resident-weights kernel for the 64 -> 64 channel 3x3 stride-1 layers (forward, lazy input, data gradient)
bool maai_conv_c64_supported(const ConvArgs& a, int dtype);
int maai_conv_c64_rows(const ConvArgs& a);
int maai_conv_c64_launch(const ConvArgs& a, hipStream_t st);
int maai_conv_pp_launch(const ConvArgs& a, hipStream_t st);
