// 8-wave ping-pong WEIGHT GRADIENT (bf16) for the MFMA-bound layers with >= 256 output channels (autograd of nn.Conv2d,
// resnet.py:20-28):   dw[co][kh][kw][ci] += sum_m dy[m][co] * x[n, oh*s-ph+kh, ow*s-pw+kw, ci]
// GEMM view: rows = Cout, columns = the flattened (tap, ci) index, contraction over the pixels m, split across workgroups and
// added with fp32 atomics.  Same pipeline as conv_pp.hip — a 512-thread workgroup owns a 256 x 256 tile of dw, walks the
// pixels in 64-deep tiles, four phases per tile, the two halves of the workgroup one barrier apart, one operand quarter
// (re)filled by LDS-DMA per phase with >= 5 phases of flight — see that file for the schedule and its hazard argument.
// What differs: both operands are stored pixel-major (NHWC), so a quarter is [64 pixels][128 channels] (256-byte rows,
// chunk ^= 2*(row & 7) swizzle on the DMA source and on the reads) and the MFMA fragments come out transposed through
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10; addressing as in conv_wgrad.hip's ring kernel).
//   A_q[mi] = dy columns { wr*128 + mi*64 + r }  at position wr*64 + r        (wave rows: output channels)
//   B_q[ni] = x  columns { wc*64 + ni*32 + c }   at position wc*32 + c        (wave columns: (tap, ci))
#ifndef MAAI_EXP
#define MAAI_EXP 0
#endif
#include "common.h"
#include "maai_internal.h"
#include "conv_ppw.h"
#include <type_traits>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_w;
static __device__ uint4 g_pzero64[4];

__device__ __forceinline__ void pdma16(const void* gsrc, char* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void pw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(512, 2) void wgrad_pp_kernel(PpwArgs a) {
  typedef bf16_t T;
  constexpr int WGN = 4;
  constexpr int AQ = 16384, BQ = 16384, ABUF = 2 * AQ, BBUF = 2 * BQ, AREG = 0, BREG = 2 * ABUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WGN, wc = wid % WGN;
  const int KN = a.KH * a.KW * a.Cin;
  // XCD-aware decode (speed only): the tiles of one pixel split get ids that differ by multiples of 8 and share an XCD's L2
  const int ntiles = a.nCoB * a.nKB;
  const int bid = blockIdx.x;
  const int split_id = (bid & 7) + 8 * (bid / (8 * ntiles));
  const int tile_id = (bid >> 3) % ntiles;
  const int kb = tile_id % a.nKB, cob = tile_id / a.nKB;
  const int co0 = cob * 256, kk0 = kb * 256;
  const long long ps64 = (long long)split_id * a.pix_per_split;
  if (ps64 >= a.M) return;   // padding block of the last group of 8 splits (before any barrier)
  const int ps = (int)ps64;
  int pe = ps + a.pix_per_split;
  if (pe > a.M) pe = a.M;
  int KT = (pe - ps + 63) >> 6;
  if (KT < 2) KT = 2;         // (pixels past `pe` read the zero page: an all-zero second tile)
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const T* zsrc = reinterpret_cast<const T*>(g_pzero64);
  const int ohw = a.OH * a.OW;
  const bool dense = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0);

  // ---- LDS-DMA roles: instruction i of a quarter covers pixel rows i*32 + wid*4 + (lane >> 4), physical chunk lane & 15 ----
  const int prow = wid * 4 + (lane >> 4);
  const int lchunk = (lane & 15) ^ ((prow & 7) << 1);      // logical 8-channel chunk of the quarter's 128 columns
  int yoff[2];                                             // [mi]: dy column of this lane's chunk
  int xtoff[2], xkh[2], xkw[2];                            // [ni]: x element offset (kh*IW + kw)*Cin + ci of its chunk; its tap
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pos = lchunk * 8;
    yoff[q] = co0 + (pos >> 6) * 128 + q * 64 + (pos & 63);
    const int kk = kk0 + (pos >> 5) * 64 + q * 32 + (pos & 31);
    const int tap = kk / a.Cin, ci = kk - tap * a.Cin;
    xkh[q] = tap / a.KW;
    xkw[q] = tap - xkh[q] * a.KW;
    xtoff[q] = (xkh[q] * a.IW + xkw[q]) * a.Cin + ci;
  }
  // pixel state of this lane's two rows (tile 0): index, and for spatial layers (n, oh, ow) -> input origin offset
  int pm[2], pih[2], piw[2], poh[2], pow_[2], pbase[2];
  const int row_step = a.stride * a.IW * a.Cin;          // input elements per output row
  const int img_step = a.IH * a.IW * a.Cin;              // ... per image
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = ps + prow + 32 * i;
    pm[i] = m;
    const int n = m / ohw, rem = m - n * ohw;
    poh[i] = rem / a.OW;
    pow_[i] = rem - poh[i] * a.OW;
    pih[i] = poh[i] * a.stride - a.pad_h;
    piw[i] = pow_[i] * a.stride - a.pad_w;
    pbase[i] = ((n * a.IH + pih[i]) * a.IW + piw[i]) * a.Cin;   // (host: the input has < 2^31 elements)
  }
  // 64 pixels further, without a division: whole output rows first (64 = q*OW + rr, host-side constants), then the rest
  const int adv_q = 64 / a.OW, adv_r = 64 - adv_q * a.OW;
  auto advance_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pm[i] += 64;
      if (!dense) {
        int dq = adv_q;
        pow_[i] += adv_r;
        if (pow_[i] >= a.OW) {
          pow_[i] -= a.OW;
          ++dq;
        }
        poh[i] += dq;
        int base = pbase[i] + dq * row_step + (pow_[i] * a.stride - a.pad_w - piw[i]) * a.Cin;
        while (poh[i] >= a.OH) {   // into the next image (at most a few times: OH*OW >= 64 is not required)
          poh[i] -= a.OH;
          base += img_step - a.OH * row_step;
        }
        pih[i] = poh[i] * a.stride - a.pad_h;
        piw[i] = pow_[i] * a.stride - a.pad_w;
        pbase[i] = base;
      }
    }
  };
  auto issue_a = [&](int mi, int buf) {
    char* dst = smem + AREG + buf * ABUF + mi * AQ + wid * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T* src = (pm[i] < pe) ? dy + (long long)pm[i] * a.Cout + yoff[mi] : zsrc;
      pdma16(src, dst + i * 8192);
    }
  };
  auto issue_b = [&](int ni, int buf) {
    char* dst = smem + BREG + buf * BBUF + ni * BQ + wid * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T* src = zsrc;
      if (pm[i] < pe) {
        if (dense) {
          src = x + (long long)pm[i] * a.Cin + xtoff[ni];
        } else if ((unsigned)(pih[i] + xkh[ni]) < (unsigned)a.IH && (unsigned)(piw[i] + xkw[ni]) < (unsigned)a.IW) {
          src = x + (long long)(pbase[i] + xtoff[ni]);
        }
      }
      pdma16(src, dst + i * 8192);
    }
  };
  const int NE = 4 * KT;
  auto issue_kind = [&](auto Kc, int buf) {
    constexpr int kind = decltype(Kc)::value;
    if constexpr (kind == 0) issue_a(0, buf);
    else if constexpr (kind == 1) issue_b(0, buf);
    else if constexpr (kind == 2) issue_b(1, buf);
    else {
      issue_a(1, buf);
      advance_tile();
    }
  };

  // ---- transposed fragment reads: lane (q = li >> 2, p = li & 3) of lane group fg addresses pixel row 4*fg + q (+16, +32 s),
  //      columns cb + 4p .. +3 of a 16-column tile ----
  const int li = lane & 15, fg = lane >> 4;
  const int q4 = li >> 2, p4 = li & 3;
  const int r_lo = 4 * fg + q4;
  const int rsw = (r_lo & 7) << 1;
  const char* a_rd[4];
  const char* b_rd[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_rd[i] = smem + AREG + r_lo * 256 + (((wr * 8 + i * 2 + (p4 >> 1)) ^ rsw) << 4) + ((p4 & 1) << 3);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_rd[j] = smem + BREG + r_lo * 256 + (((wc * 4 + j * 2 + (p4 >> 1)) ^ rsw) << 4) + ((p4 & 1) << 3);
  auto tr_frag = [&](const char* base) {   // pixels 4fg+q .. of this 32-pixel substep: rows base, base + 16 rows
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_w*)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_w*)(base + 16 * 256));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bf0[2][2], bf1[2][2];

  // ---- prologue: six quarters in flight, the first three landed; waves 4-7 fall one barrier behind ----
  issue_kind(std::integral_constant<int, 0>(), 0);   // A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
  issue_kind(std::integral_constant<int, 1>(), 0);
  issue_kind(std::integral_constant<int, 2>(), 0);
  issue_kind(std::integral_constant<int, 3>(), 0);
  issue_kind(std::integral_constant<int, 0>(), 1);
  issue_kind(std::integral_constant<int, 1>(), 1);
  pw_wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  if (wid >= 4) __builtin_amdgcn_s_barrier();   // (halves by wave id: waves 4-7 are the second wave of every SIMD, whatever the wave grid)

  auto tail_wait = [&](int g) {
    int allowed = 0;
    for (int e = g + 3; e < g + 6 && e < NE; ++e) allowed += 2;
    switch (allowed) {
      case 0: pw_wait_vm<0>(); break;
      case 2: pw_wait_vm<2>(); break;
      case 4: pw_wait_vm<4>(); break;
      default: pw_wait_vm<6>(); break;
    }
  };
  auto phase = [&](auto Pc, auto Xc, auto Sc, int kt) {
    constexpr int P = decltype(Pc)::value, X = decltype(Xc)::value;
    constexpr bool STEADY = decltype(Sc)::value;
    constexpr int MI = (P >= 2) ? 1 : 0, NI = (P == 1 || P == 2) ? 1 : 0;
    if constexpr (P == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf0[j][s] = tr_frag(b_rd[j] + X * BBUF + s * 8192);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = tr_frag(a_rd[i] + X * ABUF + s * 8192);
    } else if constexpr (P == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf1[j][s] = tr_frag(b_rd[j] + X * BBUF + BQ + s * 8192);
    } else if constexpr (P == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = tr_frag(a_rd[i] + X * ABUF + AQ + s * 8192);
    }
    constexpr int KIND = (P + 2) & 3;
    constexpr int BUF = (P < 2) ? (X ^ 1) : X;
    bool do_issue = true;
    if constexpr (STEADY) {
      pw_wait_vm<6>();
    } else {
      const int g = 4 * kt + P;
      if (g + 6 <= NE) pw_wait_vm<6>(); else tail_wait(g);
      do_issue = g + 6 < NE;
    }
#if !(MAAI_EXP & 4)
    if (STEADY || do_issue) issue_kind(std::integral_constant<int, KIND>(), BUF);
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    auto mma_rows = [&](auto I0, auto I1) {
#pragma unroll
      for (int i = decltype(I0)::value; i < decltype(I1)::value; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if constexpr (NI == 0)
              acc[MI * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf0[j][s], acc[MI * 4 + i][j], 0, 0, 0);
            else
              acc[MI * 4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf1[j][s], acc[MI * 4 + i][2 + j], 0, 0, 0);
          }
    };
    // (MAAI_EXP & 4: the A/B placement of conv_pp.hip, measured slower)
#if !(MAAI_EXP & 4)
    mma_rows(std::integral_constant<int, 0>(), std::integral_constant<int, 4>());
#else
    mma_rows(std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
    __builtin_amdgcn_sched_barrier(0);
    if (STEADY || do_issue) issue_kind(std::integral_constant<int, KIND>(), BUF);
    __builtin_amdgcn_sched_barrier(0);
    mma_rows(std::integral_constant<int, 1>(), std::integral_constant<int, 4>());
#endif
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto tile = [&](auto Xc, auto Sc, int kt) {
    phase(std::integral_constant<int, 0>(), Xc, Sc, kt);
    phase(std::integral_constant<int, 1>(), Xc, Sc, kt);
    phase(std::integral_constant<int, 2>(), Xc, Sc, kt);
    phase(std::integral_constant<int, 3>(), Xc, Sc, kt);
  };
  int kt = 0;
  for (; kt + 3 < KT; kt += 2) {
    tile(std::integral_constant<int, 0>(), std::true_type(), kt);
    tile(std::integral_constant<int, 1>(), std::true_type(), kt + 1);
  }
  tile(std::integral_constant<int, 0>(), std::false_type(), kt);
  if (kt + 1 < KT) tile(std::integral_constant<int, 1>(), std::false_type(), kt + 1);
  if (kt + 2 < KT) tile(std::integral_constant<int, 0>(), std::false_type(), kt + 2);
  if (wid < 4) __builtin_amdgcn_s_barrier();

  // ---- epilogue: fp32 atomics; C layout row (co) = 4*fg + r, column (tap, ci) = li ----
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wr * 128 + i * 16 + 4 * fg + r;
        const int kk = kk0 + wc * 64 + j * 16 + li;
        atomicAdd(a.dw + (long long)co * KN + kk, acc[i][j][r]);
      }
}

// Shapes: bf16, Cout % 256 == 0, (KH*KW*Cin) % 256 == 0, Cin % 32 == 0 (a 32-column wave quarter never straddles a tap),
// inputs of fewer than 2^31 elements.  `target`: workgroups to aim for (<= 0: one per CU).
bool maai_wgrad_pp_supported(const PpwArgs& a) {
  const long long KN = (long long)a.KH * a.KW * a.Cin;
  return a.Cout % 256 == 0 && KN % 256 == 0 && a.Cin % 32 == 0 && (long long)a.N * a.IH * a.IW * a.Cin < (1ll << 31) &&
         (long long)a.M * a.Cout < (1ll << 40) && a.M >= 4096;
}

int maai_wgrad_pp_launch(PpwArgs a, hipStream_t st, int target) {
  const int KN = a.KH * a.KW * a.Cin;
  a.nCoB = a.Cout / 256;
  a.nKB = KN / 256;
  const long long tiles = (long long)a.nCoB * a.nKB;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (target <= 0) target = cus;           // one 8-wave workgroup per CU, one round
  const int ktiles = (a.M + 63) / 64;
  long long split = (target + tiles - 1) / tiles;
  if (split > ktiles / 8) split = ktiles / 8;
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  const int tiles_per = (int)((ktiles + split - 1) / split);
  a.pix_per_split = tiles_per * 64;
  const int ny = (a.M + a.pix_per_split - 1) / a.pix_per_split;
  const int ny8 = (ny + 7) / 8 * 8;
  constexpr int lds = 131072;
  static int attr[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&wgrad_pp_kernel), lds, attr);
  MAAI_NOTE_KERNEL(wgrad_pp_kernel);
  hipLaunchKernelGGL(wgrad_pp_kernel, dim3((unsigned)(tiles * ny8)), dim3(512), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}
