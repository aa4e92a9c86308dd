// Weight gradient of the implicit-GEMM convolution (autograd of nn.Conv2d /
// nn.Linear on the path: resnet.py:20-28,169; multilayerPerceptron.py:12-16).
//
//   dw[co][kh][kw][ci] += sum_m dy[m][co] * x[n, oh*s-ph+kh, ow*s-pw+kw, ci]
//
// GEMM view: rows = Cout, cols = (tap, Cin), contraction over the N*OH*OW
// pixels, split across workgroups (split-K) and added with fp32 atomics.
// Both operands are stored pixel-major (NHWC), i.e. the contraction index is
// the SLOW index of both — the MFMA wants it fastest.  bf16: tiles are staged
// pixel-major in LDS and the fragments come out transposed through
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): lane group g, read h, row q
// <-> pixel 16h+4g+q, so one wave-instruction's 32 lanes cover 8 consecutive
// pixel rows and, with a row pitch of 2*BC+32 bytes, all 64 banks exactly once.
// f32: v_mfma_f32_16x16x4_f32 takes one k per lane group, which is a plain
// ds_read_b32 of the pixel-major tile (pitch BC+16 words, conflict-free).
#include "common.h"
#include "maai_internal.h"
#include "conv_ppw.h"
#include "conv_wgrad3w.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct WgradArgs {
  const void* x;
  const void* dy;
  float* dw;
  int M;  // N*OH*OW
  int N, IH, IW, Cin, Cout, KH, KW, stride, pad_h, pad_w, OH, OW;
  int nCoB, nCiB, nTap;
  int pix_per_split;
  // XF: the x operand is act(x*xs[ci] + xt[ci]) — x is the RAW convolution output of the unit below and the weight
  // gradient needs that unit's normalised activation, which the forward pass never stored (conv_igemm.h, XF)
  const float* xs;
  const float* xt;
  int x_relu;
};

// act(v*s + t) on a 16-byte chunk, with the arithmetic (and rounding) of maai_bn_act_fwd (common.h, XfMath)
template <typename T>
__device__ __forceinline__ void xf_apply(Vec16<T>& v, const float* cs, const float* ct, int relu) {
  XfMath<T>::template run<false>(v, v, cs, ct, nullptr, nullptr, relu, false);
}

template <bool ASM>
__device__ __forceinline__ void wdma16(const void* gsrc, char* lds_dst) {
  if constexpr (ASM) {  // invisible to hipcc, which would order every LDS access behind all LDS-DMA in flight (conv_igemm.h, dma16)
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

template <typename T, int BCO, int BCI, bool XF = false>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int PK = BF ? 32 : 16;  // pixels per K-step
  constexpr int PITCH_Y = BF ? (2 * BCO + 32) : (4 * (BCO + 16));  // bytes
  constexpr int PITCH_X = BF ? (2 * BCI + 32) : (4 * (BCI + 16));
  constexpr int TILE_Y = PK * PITCH_Y, TILE_X = PK * PITCH_X, STAGE = TILE_Y + TILE_X;
  constexpr int CPRY = BCO / EPC, CPRX = BCI / EPC;
  constexpr int NLY = (PK * CPRY + 255) / 256, NLX = (PK * CPRX + 255) / 256;
  constexpr int WCO = BCO / 2, WCI = BCI / 2, TM = WCO / 16, TN = WCI / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  int id = blockIdx.x;
  const int tap = id % a.nTap;
  id /= a.nTap;
  const int cib = id % a.nCiB, cob = id / a.nCiB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co0 = cob * BCO, ci0 = cib * BCI;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const int ps = blockIdx.y * a.pix_per_split;
  int pe = ps + a.pix_per_split;
  if (pe > a.M) pe = a.M;
  const int ohw = a.OH * a.OW;

  uint4 ry[NLY], rx[NLX];
  auto load_tile = [&](int p0) {
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPRY, ch = idx - row * CPRY;
      const int m = p0 + row;
      ry[i] = (row < PK && m < pe) ? *reinterpret_cast<const uint4*>(dy + (long long)m * a.Cout + co0 + ch * EPC)
                                   : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPRX, ch = idx - row * CPRX;
      const int m = p0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < PK && m < pe) {
        const int n = m / ohw, rem = m - n * ohw;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        const int ih = oh * a.stride - a.pad_h + kh, iw = ow * a.stride - a.pad_w + kw;
        if ((unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW) {
          v = *reinterpret_cast<const uint4*>(x + (((long long)n * a.IH + ih) * a.IW + iw) * a.Cin + ci0 + ch * EPC);
          if constexpr (XF) {  // padding stays zero: the convolution pads the activation
            Vec16<T> t;
            t.raw = __builtin_bit_cast(decltype(t.raw), v);
            xf_apply<T>(t, a.xs + ci0 + ch * EPC, a.xt + ci0 + ch * EPC, a.x_relu);
            v = __builtin_bit_cast(uint4, t.raw);
          }
        }
      }
      rx[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    char* sy = smem + buf * STAGE;
    char* sx = sy + TILE_Y;
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPRY, ch = idx - row * CPRY;
      if (row < PK) *reinterpret_cast<uint4*>(sy + row * PITCH_Y + ch * 16) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPRX, ch = idx - row * CPRX;
      if (row < PK) *reinterpret_cast<uint4*>(sx + row * PITCH_X + ch * 16) = rx[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, g = lane >> 4;
  const int nsteps = (pe - ps + PK - 1) / PK;
  if (nsteps > 0) {
    load_tile(ps);
    store_tile(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_tile(ps + (s + 1) * PK);
    const char* sy = smem + buf * STAGE;
    const char* sx = sy + TILE_Y;
    if constexpr (BF) {
      // transposed fragment reads: lane (q = li>>2, p = li&3) addresses row 16h+4g+q, columns cb+4p..cb+4p+3
      bf16x8 af[TM], bfr[TN];
      const int q = li >> 2, p = li & 3;
      const int rowoff0 = (4 * g + q);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int cb = wm * WCO + i * 16;
        const char* base = sy + rowoff0 * PITCH_Y + (cb + 4 * p) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base + 16 * PITCH_Y));
        af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cb = wn * WCI + j * 16;
        const char* base = sx + rowoff0 * PITCH_X + (cb + 4 * p) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base + 16 * PITCH_X));
        bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < PK / 4; ++ks) {
        float af[TM], bfr[TN];
        const int prow = 4 * ks + g;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float*>(sy + prow * PITCH_Y + (wm * WCO + i * 16 + li) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const float*>(sx + prow * PITCH_X + (wn * WCI + j * 16 + li) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (nsteps <= 0) return;
  // epilogue: fp32 atomics; C layout row(co) = 4g + r, col(ci) = li
  const int ntap = a.KH * a.KW;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * WCO + i * 16 + 4 * g + r;
        const int ci = ci0 + wn * WCI + j * 16 + li;
        atomicAdd(a.dw + ((long long)co * ntap + tap) * a.Cin + ci, acc[i][j][r]);
      }
}


// ---------------------------------------------------------------------------
// bf16 production kernel: LDS-DMA ring (global_load_lds_dwordx4, 3 slots, counted
// vmcnt, one raw s_barrier per 32-pixel K-step).  Columns are the FLATTENED
// (tap, ci) index n' in [0, KH*KW*Cin), so narrow layers (Cin = 64, the 7x1 stem)
// still fill a 128-wide tile; dw[co][n'] is the output address as is.
// The DMA destination is lane-linear, so tiles are unpadded [32 pixels][BC] with a
// 16-byte-chunk XOR swizzle applied to the per-lane SOURCE and to the transposed
// ds_read_b64_tr_b16 reads: chunk' = chunk ^ 2*(row&7) (256-B rows) or
// chunk ^ 2*((row>>1)&3) (128-B rows) -> the 32 lanes of one read hit 32 distinct
// 8-byte slots of the 256-B bank row.
// ---------------------------------------------------------------------------
__device__ uint4 g_wzero64[4];

template <int BC>
__device__ __forceinline__ int swz(int row) {
  return BC >= 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
}

template <int BCO, int BCN, int NWM, int NWN, bool XF = false>
__global__ __launch_bounds__(64 * NWM * NWN) void wgrad_ring_kernel(WgradArgs a) {
  constexpr int PK = 32, NSTAGE = 3, NT = 64 * NWM * NWN;
  constexpr int RBY = BCO * 2, RBX = BCN * 2;           // row bytes
  constexpr int TILE_Y = PK * RBY, TILE_X = PK * RBX, STAGE = TILE_Y + TILE_X;
  constexpr int CPRY = BCO / 8, CPRX = BCN / 8;         // 16-byte chunks per row
  constexpr int NLY = PK * CPRY / NT, NLX = PK * CPRX / NT;
  constexpr int NL = NLY + NLX;
  constexpr int WCO = BCO / NWM, WCN = BCN / NWN, TM = WCO / 16, TN = WCN / 16;
  static_assert(NLY >= 1 && NLX >= 1, "tile too small for the thread count");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int wm = wid / NWN, wn = wid % NWN;
  const int KN = a.KH * a.KW * a.Cin;
  // XCD-aware decode (speed only): blocks b and b+8 share an XCD, so all tiles of one pixel split get
  // ids that differ by multiples of 8 -> they run on ONE XCD and share its L2 copy of the dY / x panels.
  const int ntiles = a.nCoB * a.nCiB;
  const int bid = blockIdx.x;
  const int split_id = (bid & 7) + 8 * (bid / (8 * ntiles));
  const int tile_id = (bid >> 3) % ntiles;
  const int cnb = tile_id % a.nCiB, cob = tile_id / a.nCiB;
  const int co0 = cob * BCO, cn0 = cnb * BCN;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dy);
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_wzero64);
  const long long ps64 = (long long)split_id * a.pix_per_split;
  if (ps64 >= a.M) return;  // padding block of the last group of 8 splits
  const int ps = (int)ps64;
  int pe = ps + a.pix_per_split;
  if (pe > a.M) pe = a.M;
  const int ohw = a.OH * a.OW;
  const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0);

  // ---- per-thread DMA slots (fixed): dY ----
  int yrow[NLY];
  long long yoff[NLY];
#pragma unroll
  for (int i = 0; i < NLY; ++i) {
    const int sidx = tid + NT * i;
    yrow[i] = sidx / CPRY;
    const int ch = (sidx % CPRY) ^ swz<BCO>(yrow[i]);
    yoff[i] = co0 + ch * 8;
  }
  // ---- X: column -> (tap, ci) fixed per slot; pixel (n, oh, ow) advances by PK per stage ----
  int xrow[NLX], xkh[NLX], xkw[NLX], xci[NLX], pn[NLX], poh[NLX], pow_[NLX];
  bool xcol_ok[NLX];
#pragma unroll
  for (int i = 0; i < NLX; ++i) {
    const int sidx = tid + NT * i;
    xrow[i] = sidx / CPRX;
    const int ch = (sidx % CPRX) ^ swz<BCN>(xrow[i]);
    const int ncol = cn0 + ch * 8;
    xcol_ok[i] = ncol < KN;
    const int tap = ncol / a.Cin;
    xci[i] = ncol - tap * a.Cin;
    xkh[i] = tap / a.KW;
    xkw[i] = tap - xkh[i] * a.KW;
    const int m = ps + xrow[i];
    pn[i] = m / ohw;
    const int rem = m - pn[i] * ohw;
    poh[i] = rem / a.OW;
    pow_[i] = rem - poh[i] * a.OW;
  }
  // XF: per-slot channel coefficients (fixed for the whole kernel) and, per ring slot, which of this thread's x chunks
  // hold real data (bit slot*NLX + i): out-of-range rows / taps / columns source the zero page and must stay zero
  float xcs[XF ? NLX : 1][8], xct[XF ? NLX : 1][8];
  unsigned okbits = 0;
  if constexpr (XF) {
#pragma unroll
    for (int i = 0; i < NLX; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xcs[i][e] = xcol_ok[i] ? a.xs[xci[i] + e] : 0.f;
        xct[i][e] = xcol_ok[i] ? a.xt[xci[i] + e] : 0.f;
      }
  }

  auto issue_stage = [&](int p0, int slot) {
    char* sy = smem + slot * STAGE + widu * 1024;
    char* sx = smem + slot * STAGE + TILE_Y + widu * 1024;
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
      const int m = p0 + yrow[i];
      const bf16_t* src = (m < pe) ? dy + (long long)m * a.Cout + yoff[i] : zsrc;
      wdma16<XF>(src, sy + i * (NT * 16));
    }
    if constexpr (XF) okbits &= ~(((1u << NLX) - 1u) << (slot * NLX));
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int m = p0 + xrow[i];
      const bf16_t* src = zsrc;
      if (m < pe && xcol_ok[i]) {
        if (dense) {
          src = x + (long long)m * a.Cin + xci[i];
        } else {
          const int ih = poh[i] * a.stride - a.pad_h + xkh[i], iw = pow_[i] * a.stride - a.pad_w + xkw[i];
          if ((unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
            src = x + (((long long)pn[i] * a.IH + ih) * a.IW + iw) * a.Cin + xci[i];
        }
      }
      if constexpr (XF) okbits |= (src != zsrc ? 1u : 0u) << (slot * NLX + i);
      wdma16<XF>(src, sx + i * (NT * 16));
      if (!dense) {  // advance this slot's pixel by PK
        pow_[i] += PK;
        while (pow_[i] >= a.OW) {
          pow_[i] -= a.OW;
          if (++poh[i] == a.OH) { poh[i] = 0; ++pn[i]; }
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, g = lane >> 4;
  const int q = li >> 2, p = li & 3;
  const int nsteps = (pe - ps + PK - 1) / PK;
  if (nsteps <= 0) return;
  const int pre = nsteps < NSTAGE - 1 ? nsteps : NSTAGE - 1;
  for (int s = 0; s < pre; ++s) issue_stage(ps + s * PK, s);
  // XF: a thread rewrites, in place, the x chunks its own LDS-DMA instructions brought (conv_igemm.h, XF): after its
  // own counted vmcnt, before the barrier that publishes the stage
  auto xform = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NLX; ++i)
      if ((okbits >> (slot * NLX + i)) & 1u) {
        bf16_t* p = reinterpret_cast<bf16_t*>(smem + slot * STAGE + TILE_Y + i * (NT * 16) + tid * 16);
        Vec16<bf16_t> v;
        v.load(p);
        xf_apply<bf16_t>(v, xcs[i], xct[i], a.x_relu);
        v.store(p);
      }
  };
  if constexpr (XF) {
    if (pre == 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    xform(0);
  }

  // transposed-read byte offsets (row 4g+q and 16+4g+q), independent of the K-step
  const int r_lo = 4 * g + q, r_hi = 16 + 4 * g + q;
  for (int s = 0; s < nsteps; ++s) {
    if constexpr (XF) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage s was retired and transformed at the end of the previous step
    } else if (nsteps - 1 - s >= 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (s + NSTAGE - 1 < nsteps) issue_stage(ps + (s + NSTAGE - 1) * PK, (s + NSTAGE - 1) % NSTAGE);
    const char* sy = smem + (s % NSTAGE) * STAGE;
    const char* sx = sy + TILE_Y;
    bf16x8 af[TM], bfr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int c = (wm * WCO + i * 16) / 8 + (p >> 1);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sy + r_lo * RBY + ((c ^ swz<BCO>(r_lo)) << 4) + ((p & 1) << 3)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sy + r_hi * RBY + ((c ^ swz<BCO>(r_hi)) << 4) + ((p & 1) << 3)));
      af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = (wn * WCN + j * 16) / 8 + (p >> 1);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sx + r_lo * RBX + ((c ^ swz<BCN>(r_lo)) << 4) + ((p & 1) << 3)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sx + r_hi * RBX + ((c ^ swz<BCN>(r_hi)) << 4) + ((p & 1) << 3)));
      bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if constexpr (XF) {
      if (s + 1 < nsteps) {  // behind this step's MFMAs: retire stage s+1 (stage s+2 may stay in flight), transform it
        if (nsteps - 2 - s >= 1) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        xform((s + 1) % NSTAGE);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * WCO + i * 16 + 4 * g + r;
        const int nc = cn0 + wn * WCN + j * 16 + li;
        if (nc < KN) atomicAdd(a.dw + (long long)co * KN + nc, acc[i][j][r]);
      }
}

template <int BCO, int BCN, int NWM, int NWN>
static int launch_wgrad_ring(WgradArgs a, hipStream_t st, int target) {
  constexpr int PK = 32, NSTAGE = 3, NT = 64 * NWM * NWN;
  constexpr int lds = NSTAGE * PK * (BCO * 2 + BCN * 2);
  const int KN = a.KH * a.KW * a.Cin;
  a.nCoB = a.Cout / BCO;
  a.nCiB = (KN + BCN - 1) / BCN;
  a.nTap = 1;
  const long long tiles = (long long)a.nCoB * a.nCiB;
  const int ksteps = (a.M + PK - 1) / PK;
  // split-K count: every split adds Cout*KN*4 bytes of fp32 atomics, which run at ~1.3 TB/s chip-wide
  // (MI355X_MICROARCH.md "Global float atomics"); more workgroups hide latency better.  The host side
  // measures a few budgets per shape and caches the best (kernels.py).
  if (target <= 0) target = 1536;
  if (NT == 512) target = (target + 1) / 2;  // one 8-wave workgroup per CU
  long long split = (target + tiles - 1) / tiles;
  if (split > ksteps / 8) split = ksteps / 8;
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  const int steps_per = (int)((ksteps + split - 1) / split);
  a.pix_per_split = steps_per * PK;
  const int ny = (a.M + a.pix_per_split - 1) / a.pix_per_split;
  const int ny8 = (ny + 7) / 8 * 8;
  static int attr_a[64] = {0}, attr_b[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&wgrad_ring_kernel<BCO, BCN, NWM, NWN, false>), lds, attr_a);
  maai_ensure_lds(reinterpret_cast<const void*>(&wgrad_ring_kernel<BCO, BCN, NWM, NWN, true>), lds, attr_b);
  if (a.xs) {
    MAAI_NOTE_KERNEL(wgrad_ring_kernel<BCO, BCN, NWM, NWN, true>);
    hipLaunchKernelGGL((wgrad_ring_kernel<BCO, BCN, NWM, NWN, true>), dim3((unsigned)(tiles * ny8)), dim3(NT), lds, st, a);
  }
  else {
    MAAI_NOTE_KERNEL(wgrad_ring_kernel<BCO, BCN, NWM, NWN, false>);
    hipLaunchKernelGGL((wgrad_ring_kernel<BCO, BCN, NWM, NWN, false>), dim3((unsigned)(tiles * ny8)), dim3(NT), lds, st, a);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// 3x3 stride-1 same-size layers, bf16: patch-staged weight gradient.
// The ring kernel above re-stages, for every (tap, ci) column tile, dY and a shifted copy of x: 3 LDS-DMA
// instructions per 8 MFMAs per wave, 5-9x the algorithmic bytes through L2 -> 360-620 TFLOP/s.  Here a workgroup
// (6 waves) owns a 64(co) x 9(taps) x 64(ci) block of dW and walks 8x16 patches of output pixels: per patch it
// stages dY [128 px][64 co] and the 10x18 input halo [10 x 24 slots][64 ci] ONCE (46 KB, 46 DMA instructions)
// and every tap reads the halo at a shifted pixel address with transposed reads — 576 MFMAs per 46 DMA
// instructions.  Wave (h, kh): output-channel half h (two 16-row tiles), kernel row kh, all three kw, all four ci
// tiles = 24 accumulator tiles.  K-step = two patch rows (32 pixels): row r of the step is pixel (2s + r/16, r%16).
// 128-byte rows, swizzle chunk ^ 2*((row>>1)&3): eight consecutive rows fill the 256-byte bank row for
// ds_read_b64_tr_b16 at ANY starting row, and 24-slot halo rows keep it a function of (x + kw) alone.
// Single-buffered (two workgroups per CU overlap each other's load and multiply phases); fp32 atomics at the end.
// ---------------------------------------------------------------------------
struct WgradPatchArgs {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, Cin, Cout;
  int tilesX, tilesY;
  int nCoB, nCiB;
  long long npatch, per_split;
  const float* xs;  // XF (see WgradArgs)
  const float* xt;
  int x_relu;
};

// DB: two patch buffers (92 KB, one workgroup per CU): the next patch's LDS-DMA is issued right after the barrier that
// opens the multiply phase of the current one, and lands (and, XF, is normalised by its owner lanes) behind its MFMAs —
// one barrier per patch instead of two, no exposed load latency.  The DMA is inline assembly there: hipcc would order
// the fragment reads behind every LDS-DMA in flight.
template <bool XF, bool DB>
__global__ __launch_bounds__(384, DB ? 2 : 3) void wgrad3x3_patch_kernel(WgradPatchArgs a) {
  constexpr int TH = 8, HW = 24, RB = 128;            // patch rows, halo slots per row, bytes per pixel row (64 ch)
  constexpr int YB = TH * 16 * RB;                    // dY tile bytes (16 KB)
  constexpr int NIY = YB / 1024, NIH = (TH + 2) * HW * RB / 1024;  // wave-wide DMA instructions: 16 + 30
  constexpr int PB = YB + (TH + 2) * HW * RB;        // bytes of one patch buffer (46 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int half = widu / 3, kh = widu - half * 3;
  const int ntiles = a.nCoB * a.nCiB;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int cob = tile / a.nCiB, cib = tile - cob * a.nCiB;
  const int co0 = cob * 64, ci0 = cib * 64;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dy);
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_wzero64);
  const long long p_begin = (long long)split * a.per_split;
  long long p_end = p_begin + a.per_split;
  if (p_end > a.npatch) p_end = a.npatch;

  f32x4 acc[2][3][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, g = lane >> 4, q = li >> 2, p = li & 3;
  const int r0 = 4 * g + q;  // this lane's row inside a 16-row half step
  // transposed-read offsets: A (dY, rows linear), B (halo, three kw shifts x four ci tiles)
  // (the ci-tile index j only flips bits 5-6 of the offset: base + (swz ^ (j << 5)), two registers per shift)
  int offA[2], baseB[3], swzB[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = 2 * (2 * half + i) + (p >> 1);
    offA[i] = r0 * RB + ((c ^ (((r0 >> 1) & 3) << 1)) << 4) + ((p & 1) << 3);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int hr = r0 + k;  // slot x-index of this lane's pixel at shift kw = k (slot 0 is column -1)
    baseB[k] = hr * RB + ((p & 1) << 3);
    swzB[k] = ((p >> 1) ^ (((hr >> 1) & 3) << 1)) << 4;
  }
  const int tpi = a.tilesX * a.tilesY;
  const int drow = lane >> 3, dch = lane & 7;  // DMA: 8 lanes per 128-byte row

  // DMA sources.  Interior patches (all but the plane's border ring) need no per-lane decode: each of this wave's
  // instructions reads patch base + a per-lane offset fixed for the whole kernel (pad slots -> the zero page).
  constexpr int NDI = (NIY + NIH + 5) / 6;
  int rel[NDI];
  unsigned padmask = 0;
#pragma unroll
  for (int i = 0; i < NDI; ++i) {
    const int qi = widu + 6 * i;
    rel[i] = 0;
    if (qi < NIY) {
      const int row = qi * 8 + drow;
      rel[i] = ((row >> 4) * a.W + (row & 15)) * a.Cout + co0 + (dch ^ (((row >> 1) & 3) << 1)) * 8;
    } else if (qi < NIY + NIH) {
      const int hp = (qi - NIY) * 8 + drow;
      const int hy = hp / HW, hx = hp - hy * HW;
      rel[i] = ((hy - 1) * a.W + (hx - 1)) * a.Cin + ci0 + (dch ^ (((hp >> 1) & 3) << 1)) * 8;
      if (hx >= 18) padmask |= 1u << i;
    }
  }

  // XF: every halo chunk this lane stages carries the same 8 input channels (the swizzle depends on the lane alone);
  // their coefficients are re-read (L1) per patch rather than held across the MFMA phase, whose 168-register budget
  // has no room for them
  const int xf_c = ci0 + (dch ^ (((drow >> 1) & 3) << 1)) * 8;

  // stage patch pt (dY tile + input halo) into buffer ybuf / hbuf
  auto issue_patch = [&](long long pt, char* ybuf, char* hbuf) {
    const int n = (int)(pt / tpi);
    const int rem = (int)(pt - (long long)n * tpi);
    const int tyi = rem / a.tilesX;
    const int oy0 = tyi * TH, ox0 = (rem - tyi * a.tilesX) * 16;
    if (oy0 >= 1 && ox0 >= 1 && oy0 + TH + 1 <= a.H && ox0 + 17 <= a.W) {
      const long long pix0 = ((long long)n * a.H + oy0) * a.W + ox0;
      const bf16_t* by = dy + pix0 * a.Cout;
      const bf16_t* bx = x + pix0 * a.Cin;
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const int qi = widu + 6 * i;
        if (qi < NIY) {
          wdma16<DB>(by + rel[i], ybuf + qi * 1024);
        } else if (qi < NIY + NIH) {
          const bf16_t* src = ((padmask >> i) & 1u) ? zsrc : bx + rel[i];
          wdma16<DB>(src, hbuf + (qi - NIY) * 1024);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const int qi = widu + 6 * i;  // wave-uniform instruction index
        if (qi < NIY) {
          const int row = qi * 8 + drow;  // pixel of the patch
          const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
          const int sc = dch ^ (((row >> 1) & 3) << 1);
          const bf16_t* src = (oy < a.H && ox < a.W) ? dy + (((long long)n * a.H + oy) * a.W + ox) * a.Cout + co0 + sc * 8 : zsrc;
          wdma16<DB>(src, ybuf + qi * 1024);
        } else if (qi < NIY + NIH) {
          const int hp = (qi - NIY) * 8 + drow;  // halo slot
          const int hy = hp / HW, hx = hp - hy * HW;
          const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
          const int sc = dch ^ (((hp >> 1) & 3) << 1);
          const bool ok = hx < 18 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          const bf16_t* src = ok ? x + (((long long)n * a.H + iy) * a.W + ix) * a.Cin + ci0 + sc * 8 : zsrc;
          wdma16<DB>(src, hbuf + (qi - NIY) * 1024);
        }
      }
    }
  };
  // (the double-buffered kernel runs at two waves per SIMD: its 256-register budget holds the coefficients for good)
  float hcs[8], hct[8];
  if constexpr (XF && DB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      hcs[e] = a.xs[xf_c + e];
      hct[e] = a.xt[xf_c + e];
    }
  }
  // XF: normalise (+ activate) this lane's own halo chunks of patch pt in place; padding (zero page) stays zero
  auto xform_patch = [&](long long pt, char* hbuf) {
    const int n = (int)(pt / tpi);
    const int rem = (int)(pt - (long long)n * tpi);
    const int tyi = rem / a.tilesX;
    const int oy0 = tyi * TH, ox0 = (rem - tyi * a.tilesX) * 16;
    float xcs[8], xct[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xcs[e] = DB ? hcs[e] : a.xs[xf_c + e];
      xct[e] = DB ? hct[e] : a.xt[xf_c + e];
    }
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int qi = widu + 6 * i;
      if (qi >= NIY && qi < NIY + NIH) {
        const int hp = (qi - NIY) * 8 + drow;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
        if (hx < 18 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          bf16_t* p = reinterpret_cast<bf16_t*>(hbuf + (qi - NIY) * 1024 + lane * 16);
          Vec16<bf16_t> v;
          v.load(p);
          xf_apply<bf16_t>(v, xcs, xct, a.x_relu);
          v.store(p);
        }
      }
    }
  };

  if constexpr (DB) {
    if (p_begin < p_end) issue_patch(p_begin, smem, smem + YB);
  }
  for (long long pt = p_begin; pt < p_end; ++pt) {
    char* ybuf = smem + (DB ? (int)((pt - p_begin) & 1) * PB : 0);
    char* hbuf = ybuf + YB;
    if constexpr (DB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's chunks of patch pt have landed
      if constexpr (XF) {
        if (pt == p_begin) xform_patch(pt, hbuf);        // (later patches were normalised inside the previous multiply phase)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();  // patch pt is complete in LDS; everyone is done multiplying patch pt-1
      if (pt + 1 < p_end) {
        char* yn = smem + (int)((pt + 1 - p_begin) & 1) * PB;
        issue_patch(pt + 1, yn, yn + YB);
      }
    } else {
      __syncthreads();  // everyone is done multiplying the previous patch
      issue_patch(pt, ybuf, hbuf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (XF) xform_patch(pt, hbuf);
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < TH / 2; ++s) {
      if constexpr (XF && DB) {
        // before the last K-step: the next patch has had three steps to land; its owner lanes normalise it while the
        // other wave of the SIMD keeps the matrix pipe busy
        if (s == TH / 2 - 1 && pt + 1 < p_end) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          xform_patch(pt + 1, smem + (int)((pt + 1 - p_begin) & 1) * PB + YB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      bf16x8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* base = ybuf + s * 32 * RB + offA[i];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(base + 16 * RB));
        af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      const char* hrow = hbuf + (2 * s + kh) * (HW * RB);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        bf16x8 bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ob = baseB[k] + (swzB[k] ^ (j << 5));
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(hrow + ob));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(hrow + HW * RB + ob));
          bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][k][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the next shift's fragment reads from being hoisted (register budget)
      }
    }
  }
  if (p_begin >= p_end) return;
  // C layout: row (co) = 4g + r, column (ci) = li
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (2 * half + i) * 16 + 4 * g + r;
          const int ci = ci0 + j * 16 + li;
          atomicAdd(a.dw + (((long long)co * 3 + kh) * 3 + k) * a.Cin + ci, acc[i][k][j][r]);
        }
}

// MAAI_WGRAD_PATCH = 0 | 1 overrides the shape rule (read per call, for tests and A/B runs)
static bool wgrad_patch_applies(const WgradArgs& a) {
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_h == 1 && a.pad_w == 1 && a.IH == a.OH && a.IW == a.OW &&
        a.Cin % 64 == 0 && a.Cout % 64 == 0))
    return false;
  const char* e = getenv("MAAI_WGRAD_PATCH");
  const int forced = e ? atoi(e) : -1;
  if (forced == 0) return false;
  if (forced == 1) return true;
  const int tx = (a.OW + 15) / 16, ty = (a.OH + 7) / 8;
  const double cover = (double)a.OH * a.OW / ((double)tx * 16 * ty * 8);
  return cover >= 0.85 && (long long)a.N * tx * ty >= 2048;
}

static int launch_wgrad_patch(const WgradArgs& w, hipStream_t st, int target) {
  WgradPatchArgs a;
  a.x = w.x; a.dy = w.dy; a.dw = w.dw;
  a.xs = w.xs; a.xt = w.xt; a.x_relu = w.x_relu;
  a.N = w.N; a.H = w.OH; a.W = w.OW; a.Cin = w.Cin; a.Cout = w.Cout;
  a.tilesX = (w.OW + 15) / 16;
  a.tilesY = (w.OH + 7) / 8;
  a.nCoB = w.Cout / 64;
  a.nCiB = w.Cin / 64;
  a.npatch = (long long)w.N * a.tilesX * a.tilesY;
  const long long tiles = (long long)a.nCoB * a.nCiB;
  if (target <= 0) target = 512;  // two 6-wave workgroups per CU
  long long split = (target + tiles - 1) / tiles;
  if (split > a.npatch) split = a.npatch;
  if (split < 1) split = 1;
  a.per_split = (a.npatch + split - 1) / split;
  split = (a.npatch + a.per_split - 1) / a.per_split;
  constexpr int lds1 = 8 * 16 * 128 + 10 * 24 * 128;  // 16 KB + 30 KB per patch buffer
  const char* e = getenv("MAAI_WGRAD_PATCH_DB");         // 0: the single-buffered kernel (two workgroups per CU), for A/B runs
  const bool db = !(e && atoi(e) == 0);
  const int lds = db ? 2 * lds1 : lds1;
  static int attr[4][64] = {{0}};
  const dim3 grid((unsigned)(tiles * split));
  if (a.xs) {
    if (db) {
      maai_ensure_lds(reinterpret_cast<const void*>(&wgrad3x3_patch_kernel<true, true>), lds, attr[0]);
      MAAI_NOTE_KERNEL(wgrad3x3_patch_kernel<true, true>);
      hipLaunchKernelGGL((wgrad3x3_patch_kernel<true, true>), grid, dim3(384), lds, st, a);
    } else {
      maai_ensure_lds(reinterpret_cast<const void*>(&wgrad3x3_patch_kernel<true, false>), lds, attr[1]);
      MAAI_NOTE_KERNEL(wgrad3x3_patch_kernel<true, false>);
      hipLaunchKernelGGL((wgrad3x3_patch_kernel<true, false>), grid, dim3(384), lds, st, a);
    }
  } else {
    if (db) {
      maai_ensure_lds(reinterpret_cast<const void*>(&wgrad3x3_patch_kernel<false, true>), lds, attr[2]);
      MAAI_NOTE_KERNEL(wgrad3x3_patch_kernel<false, true>);
      hipLaunchKernelGGL((wgrad3x3_patch_kernel<false, true>), grid, dim3(384), lds, st, a);
    } else {
      maai_ensure_lds(reinterpret_cast<const void*>(&wgrad3x3_patch_kernel<false, false>), lds, attr[3]);
      MAAI_NOTE_KERNEL(wgrad3x3_patch_kernel<false, false>);
      hipLaunchKernelGGL((wgrad3x3_patch_kernel<false, false>), grid, dim3(384), lds, st, a);
    }
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// Tile choice: every (co-tile, column-tile) pair streams all pixels, so operand traffic is
// M * (BCO + BCN) * 2 B * (Cout/BCO) * (KN/BCN).  Measured with FETCH_SIZE, the 128x128 tile moved 2.6x the
// algorithmic bytes (profiles/r01_pmc_traffic_b256.json) and ran at the fabric rate, not the MFMA rate;
// an 8-wave 256x256 tile halves that but, at one workgroup per CU, measured slower (scripts/conv_micro.py);
// it stays selectable with MAAI_WGRAD_TILE=256.
static int dispatch_wgrad_bf16(const WgradArgs& a, hipStream_t st, int target) {
  const int KN = a.KH * a.KW * a.Cin;
  static const int big = getenv("MAAI_WGRAD_TILE") ? atoi(getenv("MAAI_WGRAD_TILE")) : 128;  // measured: the 8-wave 256x256 tile is slower (1 workgroup per CU)
  if (big == 256 && a.Cout % 256 == 0 && KN >= 256 && a.M >= 16384) return launch_wgrad_ring<256, 256, 2, 4>(a, st, target);
  const bool co128 = a.Cout % 128 == 0;
  if (KN >= 128) return co128 ? launch_wgrad_ring<128, 128, 2, 2>(a, st, target) : launch_wgrad_ring<64, 128, 2, 2>(a, st, target);
  return co128 ? launch_wgrad_ring<128, 64, 2, 2>(a, st, target) : launch_wgrad_ring<64, 64, 2, 2>(a, st, target);
}

template <typename T, int BCO, int BCI>
static int launch_wgrad(WgradArgs a, hipStream_t st) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int PK = BF ? 32 : 16;
  constexpr int PITCH_Y = BF ? (2 * BCO + 32) : (4 * (BCO + 16));
  constexpr int PITCH_X = BF ? (2 * BCI + 32) : (4 * (BCI + 16));
  constexpr int lds = 2 * PK * (PITCH_Y + PITCH_X);
  a.nCoB = a.Cout / BCO;
  a.nCiB = a.Cin / BCI;
  a.nTap = a.KH * a.KW;
  const long long tiles = (long long)a.nCoB * a.nCiB * a.nTap;
  const int ksteps = (a.M + PK - 1) / PK;
  long long split = (2048 + tiles - 1) / tiles;
  if (split > ksteps / 4) split = ksteps / 4;
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  const int steps_per = (int)((ksteps + split - 1) / split);
  a.pix_per_split = steps_per * PK;
  const int ny = (a.M + a.pix_per_split - 1) / a.pix_per_split;
  static int attr_a[64] = {0}, attr_b[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&wgrad_kernel<T, BCO, BCI, false>), lds, attr_a);
  maai_ensure_lds(reinterpret_cast<const void*>(&wgrad_kernel<T, BCO, BCI, true>), lds, attr_b);
  if (a.xs) {
    MAAI_NOTE_KERNEL(wgrad_kernel<T, BCO, BCI, true>);
    hipLaunchKernelGGL((wgrad_kernel<T, BCO, BCI, true>), dim3((unsigned)tiles, ny), dim3(256), lds, st, a);
  }
  else {
    MAAI_NOTE_KERNEL(wgrad_kernel<T, BCO, BCI, false>);
    hipLaunchKernelGGL((wgrad_kernel<T, BCO, BCI, false>), dim3((unsigned)tiles, ny), dim3(256), lds, st, a);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

template <typename T>
static int dispatch_wgrad(const WgradArgs& a, hipStream_t st) {
  const bool co128 = a.Cout % 128 == 0;
  if (a.Cin % 128 == 0) return co128 ? launch_wgrad<T, 128, 128>(a, st) : launch_wgrad<T, 64, 128>(a, st);
  if (a.Cin % 64 == 0) return co128 ? launch_wgrad<T, 128, 64>(a, st) : launch_wgrad<T, 64, 64>(a, st);
  return co128 ? launch_wgrad<T, 128, 32>(a, st) : launch_wgrad<T, 64, 32>(a, st);
}

extern "C" int maai_conv2d_wgrad(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype, void* stream) {
  return maai_conv2d_wgrad_tuned(d, x, dy, dw, dtype, 0, stream);
}

extern "C" int maai_conv2d_wgrad_tuned(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype,
                                       int target_blocks, void* stream) {
  return maai_conv2d_wgrad_xf(d, x, dy, dw, dtype, target_blocks, nullptr, nullptr, 0, stream);
}

extern "C" int maai_conv2d_wgrad_xf(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype,
                                    int target_blocks, const float* xs, const float* xt, int x_relu, void* stream) {
  MAAI_CHECK_ARG(d && x && dy && dw, "conv2d_wgrad: null pointer");
  MAAI_CHECK_ARG((xs == nullptr) == (xt == nullptr), "conv2d_wgrad: xs and xt come together");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "conv2d_wgrad: dtype must be MAAI_BF16 or MAAI_F32");
  MAAI_CHECK_ARG(d->Cin % 32 == 0, "conv2d_wgrad: Cin must be a multiple of 32");
  MAAI_CHECK_ARG(d->Cout % 64 == 0, "conv2d_wgrad: Cout must be a multiple of 64");
  MAAI_CHECK_ARG(d->out_stride == 1 && d->out_off_h == 0 && d->out_off_w == 0 && d->OHg == d->OH && d->OWg == d->OW,
                 "conv2d_wgrad: dy must be the dense output grid");
  const long long M = (long long)d->N * d->OH * d->OW;
  MAAI_CHECK_ARG(M > 0 && M < (1ll << 31), "conv2d_wgrad: pixel count must fit 31 bits");
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.M = (int)M;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
  a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.OH = d->OH; a.OW = d->OW;
  a.nCoB = a.nCiB = a.nTap = 0; a.pix_per_split = 0;
  a.xs = xs; a.xt = xt; a.x_relu = x_relu;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MAAI_BF16 && !getenv("MAAI_WGRAD_LEGACY") && !xs && a.KH == 3 && a.KW == 3 && (a.stride == 1 || a.stride == 2) && a.pad_h == 1 &&
      a.pad_w == 1) {
    // the eight-wave wide patch kernel (conv_wgrad3w.hip).  MAAI_WGRAD_WIDE = 0 (off) | 1 (shape rule, default) | 2 (wherever built)
    const char* e = getenv("MAAI_WGRAD_WIDE");
    const int mode = e ? atoi(e) : 1;
    Wgrad3wArgs w;
    w.x = x; w.dy = dy; w.dw = dw; w.N = a.N; w.H = a.OH; w.W = a.OW; w.Cin = a.Cin; w.Cout = a.Cout;
    w.IH = a.IH; w.IW = a.IW; w.stride = a.stride;
    w.tilesX = w.tilesY = w.nCoB = w.nCiB = 0; w.npatch = w.per_split = 0;
    bool rule = false;
    if (mode != 0 && maai_wgrad3w_supported(w, &rule) && (mode == 2 || (rule && !getenv("MAAI_WGRAD_PATCH") && !getenv("MAAI_WGRAD_TILE") && !getenv("MAAI_WGRAD_PP"))))
      return maai_wgrad3w_launch(w, st, target_blocks > 0 ? (target_blocks / 6 < 64 ? 64 : target_blocks / 6) : 0);
  }
  if (dtype == MAAI_BF16 && !getenv("MAAI_WGRAD_LEGACY") && !xs) {
    // the 8-wave ping-pong kernel (conv_ppw.hip) for the MFMA-bound layers.  MAAI_WGRAD_PP = 0 (off) | 1 (shape rule,
    // default) | 2 (every shape it is built for), read per call.
    const char* e = getenv("MAAI_WGRAD_PP");
    const int mode = e ? atoi(e) : 1;
    if (mode != 0) {
      PpwArgs p;
      p.x = x; p.dy = dy; p.dw = dw; p.M = a.M;
      p.N = a.N; p.IH = a.IH; p.IW = a.IW; p.Cin = a.Cin; p.Cout = a.Cout; p.KH = a.KH; p.KW = a.KW;
      p.stride = a.stride; p.pad_h = a.pad_h; p.pad_w = a.pad_w; p.OH = a.OH; p.OW = a.OW;
      p.nCoB = p.nKB = 0; p.pix_per_split = 0;
      // measured against the ring / patch kernels (scripts/ppw_ab.py, 256 images, interleaved): the pointwise layers gain
      // 35-60 % (256->1024@56 0.69 -> 0.45 ms, 1024->512@56 1.18 -> 0.74, 512->2048@28 0.61 -> 0.40, strided shortcuts 0.38 ->
      // 0.30); the 3x3 layers do not (256@56: 1.28 ms on the patch kernel vs 2.06; 512@28 and the strided ones tie) — their
      // x operand is gathered per tap and pixel, which this kernel pays in address arithmetic
      // (512-channel 3x3 layers, which the patch kernel's shape rule leaves to the ring kernel: 1.93 -> 1.68 ms at 28^2,
      //  the stride-2 one at 56^2 2.04 -> 1.68)
      const bool rule = (a.KH * a.KW == 1 && a.Cin >= 256) || (a.KH * a.KW == 9 && a.Cin >= 512);
      if (maai_wgrad_pp_supported(p) && (mode == 2 || (rule && !getenv("MAAI_WGRAD_PATCH") && !getenv("MAAI_WGRAD_TILE"))))
        return maai_wgrad_pp_launch(p, st, target_blocks > 0 ? (target_blocks / 6 < 64 ? 64 : target_blocks / 6) : 0);
    }
  }
  if (dtype == MAAI_BF16 && !getenv("MAAI_WGRAD_LEGACY")) {
    if (wgrad_patch_applies(a)) return launch_wgrad_patch(a, st, target_blocks > 0 ? target_blocks / 3 : 0);
    return dispatch_wgrad_bf16(a, st, target_blocks);
  }
  return dtype == MAAI_BF16 ? dispatch_wgrad<bf16_t>(a, st) : dispatch_wgrad<float>(a, st);
}
