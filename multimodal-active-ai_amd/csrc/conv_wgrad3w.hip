// 3x3 stride-1 same-size WEIGHT GRADIENT (bf16), wide patch kernel: eight waves, 36 accumulator tiles per wave
// (autograd of nn.Conv2d, resnet.py:20-23):   dw[co][kh][kw][ci] += sum_m dy[m][co] * x[n, oh-1+kh, ow-1+kw, ci]
//
// conv_wgrad.hip's patch kernel (six waves, a 64(co) x 9 x 64(ci) block of dw per workgroup) stages dY and the input halo of
// a 128-pixel patch once and reads every tap from the halo at a shifted address.  Measured on the 256-channel layer at 56^2:
// 0.74 PFLOP/s, 35 % MFMA-busy (profiles/r03_pmc_wgrad3x3_c256_sq.json).  Two things hold it there: six waves on four SIMDs
// (two SIMDs carry two waves, two carry one: the workgroup runs at the pace of the loaded pair), and 28 transposed LDS reads
// per 24 MFMAs per wave and K-step — more LDS time (4 clk per read, one LDS per CU) than matrix time.
// Here the workgroup owns 128(co) x 9(taps) x 64(ci) and has EIGHT waves, two per SIMD; wave (h, c) owns output-channel half
// h (four 16-row tiles), input-channel tile c (16 channels) and ALL nine taps: 36 accumulator tiles (144 registers), and per
// K-step 4 dY fragments + 9 halo fragments = 26 transposed reads for 36 MFMAs — 0.72 LDS clocks per matrix clock instead of
// 1.17.  dY is re-read by Cin/64 workgroups, x by Cout/128 (was Cout/64).
//   LDS: two patch buffers of 62 KB — dY as two [128 px][64 co] images (one per half h) + the halo [240 slots][64 ci];
//   128-byte rows, chunk ^ 2*((row>>1)&3) swizzle (eight consecutive rows fill the 256-byte bank row for
//   ds_read_b64_tr_b16 at any starting row; halo rows of 24 / 40 slots keep it a function of the slot's x alone).
//   Patch geometry PW x (128/PW): 16 x 8 (halo 10 x 24 slots) or 32 x 4 (halo 6 x 40 slots) — whichever covers the plane
//   better (28^2: 77 % vs 87.5 %).  K-step = 32 pixels of the patch in raster order.
//   The next patch's LDS-DMA is issued right after the barrier that opens the multiply phase of the current one (inline
//   assembly: hipcc would order the fragment reads behind every LDS-DMA in flight) — one barrier per patch.
#include "common.h"
#include "maai_internal.h"
#include "conv_wgrad3w.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_3w;
static __device__ uint4 g_3wzero[4];

__device__ __forceinline__ void w3dma16(const void* gsrc, char* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// KS (layers with 64 output channels, or an odd multiple of 64): the block is 64(co) x 9 x 64(ci); wave (h, c) owns ALL four
// output-channel tiles, input-channel tile c, all nine taps — of the K-steps {2h, 2h+1} of every patch (the two wave groups
// split the pixels of a patch and both add their sums at the end).  Same 36 accumulator tiles, same 26 reads per 36 MFMAs.
// S = 2 (the stride-2 3x3 layers that open stages 2-4; PW = 16, not KS): a patch is 16 x 4 OUTPUT pixels (two K-steps), its halo
// the 9 x 33 input pixels they touch (40 slots per row); output pixel px of a patch row reads slot 2*px + kw of halo row
// 2*py + kh — the same swizzle keeps the four rows of a transposed read on distinct banks ((slot >> 1) & 3 = (q + (kw >> 1)) & 3).
template <int PW, bool KS, int S>
__global__ __launch_bounds__(512, 1) void wgrad3x3_wide_kernel(Wgrad3wArgs a) {
  static_assert(S == 1 || (S == 2 && PW == 16 && !KS), "stride 2: 16-wide patches, 128 output channels per block");
  constexpr int NPX = S == 1 ? 128 : 64;                 // output pixels per patch
  constexpr int TH = NPX / PW, RB = 128;                 // patch rows, bytes per pixel row (64 channels)
  constexpr int HR = S * (TH - 1) + 3, VC = S * (PW - 1) + 3;   // halo rows, valid halo columns
  constexpr int HWS = S == 1 ? PW + 8 : 40;              // halo slots per row
  constexpr int YI = NPX * RB, IPI = YI / 1024;          // one dY image (16 KB; S = 2: 8 KB), DMA instructions per image
  constexpr int YB = (KS ? 1 : 2) * YI, HB = HR * HWS * RB;   // dY bytes (32 KB; KS: 16 KB), halo bytes (30 KB; S = 2: 45 KB)
  constexpr int NIY = YB / 1024, NIH = HB / 1024;        // wave-wide DMA instructions: 32 (KS: 16) + 30
  constexpr int PB = YB + HB;                            // one patch buffer (62 KB; KS: 46 KB; S = 2: 61 KB)
  constexpr int NDI = (NIY + NIH + 7) / 8;
  constexpr int ROWSTEP = S == 2 ? 4 : (PW == 16 ? 2 : 1);          // halo rows per K-step
  constexpr int HI_OFF = S == 2 ? 2 * HWS * RB : (PW == 16 ? HWS * RB : 16 * RB);   // where the second 16 pixels of a K-step sit relative to the first
  static_assert(HB % 1024 == 0 && HWS % 8 == 0 && VC <= HWS, "halo geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int widu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = widu >> 2, ct = widu & 3;
  const int ntiles = a.nCoB * a.nCiB;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int cob = tile / a.nCiB, cib = tile - cob * a.nCiB;
  const int co0 = cob * (KS ? 64 : 128), ci0 = cib * 64;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(a.dy);
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_3wzero);
  const long long p_begin = (long long)split * a.per_split;
  long long p_end = p_begin + a.per_split;
  if (p_end > a.npatch) p_end = a.npatch;
  if (p_begin >= p_end) return;   // (whole workgroup, before any barrier)

  f32x4 acc[4][3][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, g = lane >> 4, q = li >> 2, p = li & 3;
  const int r0 = 4 * g + q;   // this lane's row inside a 16-row half step
  // transposed-read offsets: A (this half's dY image, rows linear), B (halo, three kw shifts, this wave's ci tile)
  int offA[4], offB[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 2 * i + (p >> 1);
    offA[i] = (KS ? 0 : half * YI) + r0 * RB + ((c ^ (((r0 >> 1) & 3) << 1)) << 4) + ((p & 1) << 3);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int hr = S * r0 + k;   // slot x-index of this lane's pixel at shift kw = k (slot 0 is input column S*ox0 - 1)
    offB[k] = YB + hr * RB + ((p & 1) << 3) + ((((p >> 1) ^ (((hr >> 1) & 3) << 1)) << 4) ^ (ct << 5));
  }
  const int tpi = a.tilesX * a.tilesY;
  const int drow = lane >> 3, dch = lane & 7;   // DMA: 8 lanes per 128-byte row

  // DMA sources.  Interior patches need no per-lane decode: each of this wave's instructions reads patch base + a per-lane
  // offset fixed for the whole kernel (pad slots -> the zero page).
  int rel[NDI];
  unsigned padmask = 0;
#pragma unroll
  for (int i = 0; i < NDI; ++i) {
    const int qi = widu + 8 * i;
    rel[i] = 0;
    if (qi < NIY) {
      const int row = (qi % IPI) * 8 + drow;   // pixel of the patch; image qi / IPI
      rel[i] = ((row / PW) * a.W + (row % PW)) * a.Cout + co0 + (qi / IPI) * 64 + (dch ^ (((row >> 1) & 3) << 1)) * 8;
    } else if (qi < NIY + NIH) {
      const int hp = (qi - NIY) * 8 + drow;
      const int hy = hp / HWS, hx = hp - hy * HWS;
      rel[i] = ((hy - 1) * a.IW + (hx - 1)) * a.Cin + ci0 + (dch ^ (((hp >> 1) & 3) << 1)) * 8;
      if (hx >= VC) padmask |= 1u << i;
    }
  }

  auto issue_patch = [&](long long pt, char* buf) {
    const int n = (int)(pt / tpi);
    const int rem = (int)(pt - (long long)n * tpi);
    const int tyi = rem / a.tilesX;
    const int oy0 = tyi * TH, ox0 = (rem - tyi * a.tilesX) * PW;
    if (oy0 >= 1 && ox0 >= 1 && oy0 + TH <= a.H && ox0 + PW <= a.W && S * oy0 - 1 + HR <= a.IH && S * ox0 - 1 + VC <= a.IW) {
      const bf16_t* by = dy + (((long long)n * a.H + oy0) * a.W + ox0) * a.Cout;
      const bf16_t* bx = x + (((long long)n * a.IH + S * oy0) * a.IW + S * ox0) * a.Cin;
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const int qi = widu + 8 * i;
        if (qi < NIY) {
          w3dma16(by + rel[i], buf + qi * 1024);
        } else if (qi < NIY + NIH) {
          const bf16_t* src = ((padmask >> i) & 1u) ? zsrc : bx + rel[i];
          w3dma16(src, buf + qi * 1024);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const int qi = widu + 8 * i;   // wave-uniform instruction index
        if (qi < NIY) {
          const int row = (qi % IPI) * 8 + drow;
          const int oy = oy0 + row / PW, ox = ox0 + row % PW;
          const int sc = dch ^ (((row >> 1) & 3) << 1);
          const bf16_t* src = (oy < a.H && ox < a.W) ? dy + (((long long)n * a.H + oy) * a.W + ox) * a.Cout + co0 + (qi / IPI) * 64 + sc * 8 : zsrc;
          w3dma16(src, buf + qi * 1024);
        } else if (qi < NIY + NIH) {
          const int hp = (qi - NIY) * 8 + drow;
          const int hy = hp / HWS, hx = hp - hy * HWS;
          const int iy = S * oy0 + hy - 1, ix = S * ox0 + hx - 1;
          const int sc = dch ^ (((hp >> 1) & 3) << 1);
          const bool ok = hx < VC && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
          const bf16_t* src = ok ? x + (((long long)n * a.IH + iy) * a.IW + ix) * a.Cin + ci0 + sc * 8 : zsrc;
          w3dma16(src, buf + qi * 1024);
        }
      }
    }
  };

  issue_patch(p_begin, smem);
  for (long long pt = p_begin; pt < p_end; ++pt) {
    const char* buf = smem + (int)((pt - p_begin) & 1) * PB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's chunks of patch pt have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // patch pt is complete in LDS; everyone is done multiplying patch pt-1
    if (pt + 1 < p_end) issue_patch(pt + 1, smem + (int)((pt + 1 - p_begin) & 1) * PB);
    // software pipeline (MAAI_W3_PIPE): the halo fragments of the next kernel row (or of the next K-step's first row) are read
    // under the MFMAs of the current one; the dY fragments of the next K-step behind its last MFMA group
    auto load_a = [&](int s, bf16x8* af) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* base = buf + s * 32 * RB + offA[i];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(base + 16 * RB));
        af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto load_b = [&](int s, int kh, bf16x8* bfr) {
      const char* hrow = buf + (ROWSTEP * s + kh) * (HWS * RB);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(hrow + offB[k]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(hrow + HI_OFF + offB[k]));
        bfr[k] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    bf16x8 af[4], bcur[3], bnxt[3];
    constexpr int NS = KS ? 2 : NPX / 32;
    const int s0 = KS ? 2 * half : 0;
    load_a(s0, af);
    load_b(s0, 0, bcur);
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      const int s = s0 + ss;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const bool more = kh < 2 || ss < NS - 1, next_a = kh == 2 && ss < NS - 1;
        if (kh < 2) load_b(s, kh + 1, bnxt);
        else if (ss < NS - 1) load_b(s + 1, 0, bnxt);
        // i-major: a dY fragment is done after three MFMAs and its successor (next K-step) is read right behind them — it is
        // needed again nine MFMAs later
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[i][kh][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bcur[k], acc[i][kh][k], 0, 0, 0);
          if (next_a) {
            const char* base = buf + (s + 1) * 32 * RB + offA[i];
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(base));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_3w*)(base + 16 * RB));
            af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
        // the order the scheduler must keep: this region's halo reads first, then MFMAs (with the dY reads between them)
        if (more) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          if (next_a) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // (register budget: nothing is hoisted across kernel rows)
#pragma unroll
        for (int k = 0; k < 3; ++k) bcur[k] = bnxt[k];
      }
    }
  }
  // C layout: row (co) = 4g + r, column (ci) = li
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (KS ? 0 : half * 64) + i * 16 + 4 * g + r;
          const int ci = ci0 + ct * 16 + li;
          atomicAdd(a.dw + (((long long)co * 3 + kh) * 3 + k) * a.Cin + ci, acc[i][kh][k][r]);
        }
}

static void w3_geometry(const Wgrad3wArgs& a, int pw, int* tx, int* ty, double* cover) {
  const int th = (a.stride == 2 ? 64 : 128) / pw;
  *tx = (a.W + pw - 1) / pw;
  *ty = (a.H + th - 1) / th;
  *cover = (double)a.H * a.W / ((double)*tx * pw * *ty * th);
}

// MAAI_WGRAD_WIDE = 0 | 1 (shape rule, default) | 2 (every shape it is built for), read per call.
// a.H, a.W: the OUTPUT plane (= the input plane at stride 1); a.IH, a.IW: the input plane.
bool maai_wgrad3w_supported(const Wgrad3wArgs& a, bool* by_rule) {
  if (!(a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.H >= 3 && a.W >= 3)) return false;
  if (a.stride == 2) {
    if (!(a.Cout % 128 == 0 && a.H == (a.IH - 1) / 2 + 1 && a.W == (a.IW - 1) / 2 + 1)) return false;
  } else if (!(a.stride == 1 && a.IH == a.H && a.IW == a.W)) {
    return false;
  }
  int tx, ty;
  double c16, c32;
  w3_geometry(a, 16, &tx, &ty, &c16);
  w3_geometry(a, 32, &tx, &ty, &c32);
  const double cover = a.stride == 2 ? c16 : (c16 > c32 ? c16 : c32);
  *by_rule = cover >= 0.85 && (long long)a.N * a.H * a.W >= 65536;
  return true;
}

int maai_wgrad3w_launch(Wgrad3wArgs a, hipStream_t st, int target) {
  int tx16, ty16, tx32, ty32;
  double c16, c32;
  w3_geometry(a, 16, &tx16, &ty16, &c16);
  w3_geometry(a, 32, &tx32, &ty32, &c32);
  int pw = c32 > c16 + 1e-9 ? 32 : 16;
  const char* e = getenv("MAAI_WGRAD_WIDE_PW");
  if (e && (atoi(e) == 16 || atoi(e) == 32)) pw = atoi(e);
  if (a.stride == 2) pw = 16;
  a.tilesX = pw == 16 ? tx16 : tx32;
  a.tilesY = pw == 16 ? ty16 : ty32;
  const bool ks = a.Cout % 128 != 0;
  a.nCoB = a.Cout / (ks ? 64 : 128);
  a.nCiB = a.Cin / 64;
  a.npatch = (long long)a.N * a.tilesX * a.tilesY;
  const long long tiles = (long long)a.nCoB * a.nCiB;
  if (target <= 0) target = 256;   // one 8-wave workgroup per CU
  long long split = (target + tiles - 1) / tiles;
  if (split > a.npatch) split = a.npatch;
  if (split < 1) split = 1;
  a.per_split = (a.npatch + split - 1) / split;
  split = (a.npatch + a.per_split - 1) / a.per_split;
  // two patch buffers: dY (two images; KS: one) + halo — 62 KB (KS: 46 KB; stride 2: 16 + 45 = 61 KB)
  const int lds = a.stride == 2 ? 2 * (2 * 64 * 128 + 9 * 40 * 128) : 2 * ((ks ? 1 : 2) * 128 * 128 + 240 * 128);
  static int attr[5][64] = {{0}};
  const dim3 grid((unsigned)(tiles * split));
#define MAAI_W3_LAUNCH(PW_, KS_, S_, slot)                                                                      \
  do {                                                                                                          \
    maai_ensure_lds(reinterpret_cast<const void*>(&wgrad3x3_wide_kernel<PW_, KS_, S_>), lds, attr[slot]);        \
    MAAI_NOTE_KERNEL(wgrad3x3_wide_kernel<PW_, KS_, S_>);                                                       \
    hipLaunchKernelGGL((wgrad3x3_wide_kernel<PW_, KS_, S_>), grid, dim3(512), lds, st, a);                      \
  } while (0)
  if (a.stride == 2) MAAI_W3_LAUNCH(16, false, 2, 4);
  else if (pw == 16 && !ks) MAAI_W3_LAUNCH(16, false, 1, 0);
  else if (pw == 16) MAAI_W3_LAUNCH(16, true, 1, 1);
  else if (!ks) MAAI_W3_LAUNCH(32, false, 1, 2);
  else MAAI_W3_LAUNCH(32, true, 1, 3);
#undef MAAI_W3_LAUNCH
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
