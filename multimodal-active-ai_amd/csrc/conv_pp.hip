// 8-wave ping-pong implicit-GEMM convolution (bf16) for the MFMA-bound layers: the 3x3 convolutions with >= 256 channels
// and the long-K pointwise layers of stages 3-4 (resnet.py:20-28, 101-109), forward and data gradient.
//
// The 4-wave ring kernel of conv_igemm.h multiplies one 32-deep K-step per barrier with every fragment read sitting
// between the barrier and the MFMAs that consume it: on the 256-row tile its MFMA pipe measured 38.8 % busy
// (profiles/r02_pmc_conv3x3_sq.json).  Here a 512-thread workgroup owns a 256 x 256 output tile (wave grid 2 x 4; 512 x 128
// with a 4 x 2 wave grid for 128-channel outputs) and walks K in 64-deep tiles, four phases per tile; the two halves of the workgroup (waves 0-3 and
// 4-7: the two waves of every SIMD) run ONE barrier apart, so that while one half issues its 16 MFMAs of a phase the other
// half reads its fragments for its next phase and issues LDS-DMA — the matrix pipe of a SIMD always has the other wave's
// MFMAs to run (cdna_hip_programming.md section 5, the 256^2 8-phase structure; MI355X_MICROARCH.md "Two waves per SIMD").
//
// Tile bookkeeping.  Wave (wr, wc) = (wid / WGN, wid % WGN) owns rows wr*128 .. +128 and columns wc*64 .. +64 of the tile:
// 8 x 4 accumulator tiles of 16 x 16.  A K-tile's operands live in LDS as four QUARTERS of 128-byte rows —
//   A_q[mi] = rows { wr*128 + mi*64 + r }    (the mi-th 64-row half of every wave row group: WGM*64 rows)
//   B_q[ni] = columns { wc*64 + ni*32 + c }   (the ni-th half of every wave column group: WGN*32 rows)
// — two buffers of them (128 KB for the 2 x 4 grid, 160 KB for 4 x 2).  Per K-tile a wave runs the quadrants (mi, ni) = (0,0) (0,1) (1,1) (1,0):
// phase 0 reads A_q[0] and B_q[0] fragments (B_q[0]'s stay in registers until phase 3), phase 1 B_q[1], phase 2 A_q[1],
// phase 3 nothing; each quarter is read in exactly one phase.  One quarter is (re)filled per phase, in the fixed stream
// order A0(k) B0(k) B1(k) A1(k) A0(k+1) ...; the element issued in phase g (g = 4*tile + phase) is stream element g + 6.
// That order gives every quarter >= 5 phases between its issue and its first read and >= 2 phases between its last read
// and its refill.  Synchronisation (per wave; H1 = waves 4-7 execute one extra barrier up front and so run one barrier
// behind H0):
//     phase g:  ds_read fragments of phase g            (data issued in phase <= g-5)
//               s_waitcnt vmcnt(6)                      (own LDS-DMA of phases <= g-4 has landed; 3 quarters stay in flight)
//               issue LDS-DMA of stream element g+6     (overwrites a quarter last read in phase <= g-2)
//               s_barrier #2g ; s_waitcnt lgkmcnt(0) ; 16 MFMAs ; s_barrier #2g+1
//   read-after-DMA: the DMA of phase g-5 was retired by its wave's wait of phase g-1, which precedes that wave's barrier
//     #2g-2; the reader passes its own barrier #2g-1 first — with the halves one barrier apart every pairing leaves at
//     least one common barrier between the wait and the read.
//   DMA-after-read: the reads of phase g-2 were drained (lgkmcnt(0)) before their wave's barrier #2g-3; the DMA is
//     issued after its wave's barrier #2g-1 — again at least one common barrier between them in every pairing.
// Operand staging: implicit GEMM by per-lane source address (a 128-byte channel run of one input pixel per row and tap;
// padding and rows past M read a zero page), LDS-DMA, XOR swizzle chunk ^= (row >> 1) & 7 on the source side and on
// the ds_read_b128 side (conflict-free for 128-byte rows: a lane group's 16 rows cover 16 distinct 16-byte slots).
// K order: tap-major, channels ascending, 32 per MFMA — the ring kernel's order: outputs are bit-identical to it.
// Epilogue: wave-private C area in LDS -> 16-byte row stores (128-byte lines); EMODE 0: + BatchNorm partial sums (one
// slab row per 256-row tile); EMODE 6: the data-gradient epilogue of conv_igemm.h (mask from the unit below's
// y*scale + shift > 0 or its 1-bit mask, BatchNorm-backward partial sums of the stored gradient).
#include "conv_igemm.h"
#include <type_traits>

template <int N>
__device__ __forceinline__ void pp_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WGM, int WGN, int EMODE>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvArgs a) {
  typedef bf16_t T;
  static_assert(WGM * WGN == 8, "eight waves");
  constexpr int BM = WGM * 128, BN = WGN * 64, WN = 64, TN = 4, TNQ = 2;
  constexpr int AQ = WGM * 64 * 128;     // bytes of an A quarter
  constexpr int BQ = WGN * 32 * 128;     // bytes of a B quarter
  constexpr int NAI = WGM;               // LDS-DMA instructions per thread per A quarter (64 rows each)
  constexpr int NBI = WGN / 2;           // ... per B quarter
  constexpr int ABUF = 2 * AQ, BBUF = 2 * BQ;
  constexpr int AREG = 0, BREG = 2 * ABUF;           // LDS regions: A buffers, then B buffers
  constexpr int LDC = WN + 8, CWB = 128 * LDC * 2;   // wave-private C area (epilogue)
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / WGN, wc = wid % WGN;
  const int logical = xcd_remap(blockIdx.x, a.nMB * a.nNB);
  const int mb = logical / a.nNB, nb = logical - mb * a.nNB;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  const int K = a.KH * a.KW * a.Cin;
  const int KT = K >> 6;                 // 64-deep K-tiles (host: KT >= 2)
  const int nck = a.Cin >> 6;            // K-tiles per tap
  const T* zsrc = reinterpret_cast<const T*>(g_zero64);

  // ---- LDS-DMA roles.  Instruction i of a quarter, wave `wid`, covers quarter rows (i*8 + wid)*8 + (lane >> 3), physical
  //      chunk lane & 7, which holds logical chunk (lane & 7) ^ ((row >> 1) & 7) ----
  const int prow = wid * 8 + (lane >> 3);            // quarter row of instruction 0 (instruction 1: + 64)
  const int lchunk = (lane & 7) ^ ((prow >> 1) & 7); // (+64 leaves (row >> 1) & 7 unchanged)
  int aoff[NAI][2];                                  // [i][mi]: element offset of the row's source at tap (0,0), channel lchunk*8
  int aoff2[NAI][2];                                 // ... in the SECOND source tensor (two-source pointwise data gradients, a.x2)
  unsigned amask[NAI][2];                            // bit t: tap t reads inside the image (0: the row is past M)
  const bool two = EMODE == 6 && a.x2 != nullptr;    // (host: pointwise, cin1 % 64 == 0)
  const int pitch1 = two ? a.cin1 : a.Cin, pitch2 = a.Cin - a.cin1, nck1 = a.cin1 >> 6;
  const unsigned ohw = (unsigned)(a.OHg * a.OWg);
#pragma unroll
  for (int i = 0; i < NAI; ++i) {
    const int p = prow + 64 * i;
    const int wrp = p >> 6, r = p & 63;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const long long m = (long long)mb * BM + wrp * 128 + mi * 64 + r;
      unsigned mask = 0;
      int base = 0;
      if (m < a.M) {
        const unsigned mu = (unsigned)m;
        const unsigned n = mu / ohw;
        const unsigned rem = mu - n * ohw;
        const unsigned oh = rem / (unsigned)a.OWg, ow = rem - oh * (unsigned)a.OWg;
        const int ih0 = (int)oh * a.stride - a.pad_h, iw0 = (int)ow * a.stride - a.pad_w;
        base = (int)((((long long)n * a.IH + ih0) * a.IW + iw0) * pitch1) + lchunk * 8;   // (host: the input has < 2^31 elements)
        for (int kh = 0; kh < a.KH; ++kh)
          for (int kw = 0; kw < a.KW; ++kw)
            if ((unsigned)(ih0 + kh) < (unsigned)a.IH && (unsigned)(iw0 + kw) < (unsigned)a.IW) mask |= 1u << (kh * a.KW + kw);
      }
      aoff[i][mi] = base;
      aoff2[i][mi] = two ? (int)(m < a.M ? m * pitch2 : 0) + lchunk * 8 : 0;
      amask[i][mi] = mask;
    }
  }
  int boff[NBI][2];                                  // [i][ni]: element offset of the weight row, K-tile 0, channel lchunk*8
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int p = prow + 64 * i;                     // quarter row (BN 128: 64 rows per quarter, one instruction)
    const int wcp = p >> 5, c = p & 31;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) boff[i][ni] = (nb * BN + wcp * 64 + ni * 32 + c) * K + lchunk * 8;
  }
  // issue state: the K-tile the next A quarter belongs to
  int it_tap = 0, it_ck = 0;
  int it_off = 0;              // element offset of that K-tile inside a row's receptive field: (kh*IW + kw)*Cin + ck*64
  int it_k = 0;                // its index (weights: + it_k*64)
  auto advance_tile = [&]() {
    ++it_k;
    if (++it_ck == nck) {
      it_ck = 0;
      ++it_tap;
      const int kh = it_tap / a.KW, kw = it_tap - kh * a.KW;
      it_off = (kh * a.IW + kw) * a.Cin;
    } else {
      it_off += 64;
    }
  };
  auto issue_a = [&](int mi, int buf) {
    char* dst = smem + AREG + buf * ABUF + mi * AQ + wid * 1024;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const bool ok = (amask[i][mi] >> it_tap) & 1u;
      const T* src = ok ? x + (long long)(aoff[i][mi] + it_off) : zsrc;
      if (two && ok && it_ck >= nck1) src = reinterpret_cast<const T*>(a.x2) + (long long)(aoff2[i][mi] + (it_ck - nck1) * 64);
      dma16<true>(src, dst + i * 8192);
    }
  };
  auto issue_b = [&](int ni, int buf) {
    char* dst = smem + BREG + buf * BBUF + ni * BQ + wid * 1024;
#pragma unroll
    for (int i = 0; i < NBI; ++i) dma16<true>(w + boff[i][ni] + it_k * 64, dst + i * 8192);
  };
  // stream element e = 4k + kind: kind 0 A0(k), 1 B0(k), 2 B1(k), 3 A1(k); buffer k & 1.  The element issued in phase P of
  // tile kt is e = 4*kt + P + 6: kind (P + 2) & 3 of tile kt + 1 (P = 0, 1) or kt + 2 (P = 2, 3) — all compile-time but kt.
  // Elements past the last tile do not exist: nothing is issued and the waits of the last phases count what really is in flight.
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

  // ---- fragment read addresses ----
  const int frow = lane & 15, fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const char* a_rd[2];
  const char* b_rd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ch = ((s * 4 + fg) ^ fsw) << 4;
    a_rd[s] = smem + AREG + (wr * 64 + frow) * 128 + ch;
    b_rd[s] = smem + BREG + (wc * 32 + frow) * 128 + ch;
  }

  f32x4 acc[8][TN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  frag_t af[4][2], bf0[TNQ][2], bf1[TNQ][2];

  // ---- prologue: six quarters in flight, the first three landed; H1 falls one barrier behind ----
  issue_kind(std::integral_constant<int, 0>(), 0);   // A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
  issue_kind(std::integral_constant<int, 1>(), 0);
  issue_kind(std::integral_constant<int, 2>(), 0);
  issue_kind(std::integral_constant<int, 3>(), 0);
  issue_kind(std::integral_constant<int, 0>(), 1);
  issue_kind(std::integral_constant<int, 1>(), 1);
  pp_wait_vm<2 * NAI + NBI>();   // elements 3, 4, 5 — A1(0), A0(1), B0(1) — may stay in flight (KT >= 2: they all exist)
  __builtin_amdgcn_s_barrier();
  if (wid >= 4) __builtin_amdgcn_s_barrier();   // (halves by wave id: waves 4-7 are the second wave of every SIMD, whatever the wave grid)

  // the wait of a phase in the last two tiles: elements g+3 .. g+5 that exist may stay in flight
  auto tail_wait = [&](int g) {
    int allowed = 0;
    for (int e = g + 3; e < g + 6 && e < NE; ++e) allowed += ((e & 3) == 0 || (e & 3) == 3) ? NAI : NBI;
    switch (allowed) {   // (waiting for fewer than allowed is always safe)
      case 0: pp_wait_vm<0>(); break;
      case 1: pp_wait_vm<1>(); break;
      case 2: pp_wait_vm<2>(); break;
      case 3: pp_wait_vm<3>(); break;
      case 4: pp_wait_vm<4>(); break;
      case 5: pp_wait_vm<5>(); break;
      case 6: pp_wait_vm<6>(); break;
      case 7: pp_wait_vm<7>(); break;
      case 8: pp_wait_vm<8>(); break;
      default: pp_wait_vm<9>(); break;
    }
  };

  // STEADY: kt + 2 < KT — every element of this tile's four phases exists (no conditionals in the loop body)
  auto phase = [&](auto Pc, auto Xc, auto Sc, int kt) {
    constexpr int P = decltype(Pc)::value, X = decltype(Xc)::value;
    constexpr bool STEADY = decltype(Sc)::value;
    constexpr int MI = (P >= 2) ? 1 : 0, NI = (P == 1 || P == 2) ? 1 : 0;
    // ---- fragment reads of this phase ----
    if constexpr (P == 0) {
#pragma unroll
      for (int j = 0; j < TNQ; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf0[j][s] = *reinterpret_cast<const frag_t*>(b_rd[s] + X * BBUF + j * 2048);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const frag_t*>(a_rd[s] + X * ABUF + i * 2048);
    } else if constexpr (P == 1) {
#pragma unroll
      for (int j = 0; j < TNQ; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) bf1[j][s] = *reinterpret_cast<const frag_t*>(b_rd[s] + X * BBUF + BQ + j * 2048);
    } else if constexpr (P == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const frag_t*>(a_rd[s] + X * ABUF + AQ + i * 2048);
    }
    // ---- retire the LDS-DMA of four phases ago, issue this phase's quarter ----
    // in flight behind the wait, by P: 0: {A1, A0, B0}  1: {A0, B0, B1}  2: {B0, B1, A1}  3: {B1, A1, A0}
    constexpr int INFL = (P == 0 || P == 3) ? 2 * NAI + NBI : NAI + 2 * NBI;
    constexpr int KIND = (P + 2) & 3;
    constexpr int BUF = (P < 2) ? (X ^ 1) : X;
    bool do_issue = true;
    if constexpr (STEADY) {
      pp_wait_vm<INFL>();
    } else {
      const int g = 4 * kt + P;
      if (g + 6 <= NE) pp_wait_vm<INFL>(); else tail_wait(g);
      do_issue = g + 6 < NE;
    }
#if !(MAAI_EXP & 4)
    if (STEADY || do_issue) issue_kind(std::integral_constant<int, KIND>(), BUF);
#endif
    // (measured neutral, +-1 %, on the 256- and 512-channel 3x3 layers: the fragment reads drained BEFORE this barrier instead
    //  of after it; one static s_setprio 1 for waves 4-7 instead of the pair around every MFMA group)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    auto mma_rows = [&](auto I0, auto I1) {
#pragma unroll
      for (int i = decltype(I0)::value; i < decltype(I1)::value; ++i)
#pragma unroll
        for (int j = 0; j < TNQ; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if constexpr (NI == 0)
              acc[MI * 4 + i][j] = Mma<T>::run(af[i][s], bf0[j][s], acc[MI * 4 + i][j]);
            else
              acc[MI * 4 + i][TNQ + j] = Mma<T>::run(af[i][s], bf1[j][s], acc[MI * 4 + i][TNQ + j]);
          }
    };
    // (MAAI_EXP & 4, A/B build: this phase's LDS-DMA issued HERE, behind the first four MFMAs, instead of in the load
    //  section ahead of the barrier.  Measured slower on every shape — 256->256 3x3 @56: 0.90 -> 0.99 ms, 512->512 @28: 0.76 ->
    //  0.86, the step 317.8 -> 320.1 ms: the address arithmetic and the DMA issue delay the wave's own MFMA stream more
    //  than they cost its partner from the load section.  The shipped kernel issues it in the load section.)
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
  for (; kt + 3 < KT; kt += 2) {   // two tiles per trip (the buffer index is a compile-time constant)
    tile(std::integral_constant<int, 0>(), std::true_type(), kt);
    tile(std::integral_constant<int, 1>(), std::true_type(), kt + 1);
  }
  // the last two or three tiles (kt is even here)
  tile(std::integral_constant<int, 0>(), std::false_type(), kt);
  if (kt + 1 < KT) tile(std::integral_constant<int, 1>(), std::false_type(), kt + 1);
  if (kt + 2 < KT) tile(std::integral_constant<int, 0>(), std::false_type(), kt + 2);
  if (wid < 4) __builtin_amdgcn_s_barrier();   // H0 catches up with H1's extra barrier
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                             // the operand buffers are free: the C areas reuse them

  // ---- epilogue: wave-private C area -> 16-byte row stores ----
  char* cw = smem + wid * CWB;
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  constexpr int CPR = WN / 8;                  // 16-byte chunks per row of the wave's tile
  constexpr int RPI = 64 / CPR;                // rows per wave-wide access
  const int crow = lane / CPR, cch = lane % CPR;
  const long long row0 = (long long)mb * BM + wr * 128;
  const int col0 = nb * BN + wc * WN;
  if constexpr (EMODE == 0) {
    if (a.stats) {
      // column sums of the fp32 accumulators over this wave's 128 rows: lanes with equal lane & 15 hold the same columns
      float* red = reinterpret_cast<float*>(smem + 8 * CWB);   // [WGM wr][2][BN]
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1 += acc[i][j][r];
            s2 += acc[i][j][r] * acc[i][j][r];
          }
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lane < 16) {
          red[(wr * 2 + 0) * BN + wc * WN + j * 16 + lane] = s1;
          red[(wr * 2 + 1) * BN + wc * WN + j * 16 + lane] = s2;
        }
      }
    }
  }
  {
    const uint32_t cwa = (uint32_t)(uintptr_t)(cw + ((fg * 4) * LDC + frow) * 2);
    float cbias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cbias[j] = (EMODE == 6 && a.ebias) ? a.ebias[col0 + j * 16 + frow] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[i][j];
        if constexpr (EMODE == 6) v += (f32x4){cbias[j], cbias[j], cbias[j], cbias[j]};
        const uint32_t p01 = pack_bf16x2(v[0], v[1]);
        const uint32_t p23 = pack_bf16x2(v[2], v[3]);
        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p01),
                     "n"((i * 16 * LDC + j * 16) * 2), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 2)
                     : "memory");
        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p23),
                     "n"((i * 16 * LDC + j * 16) * 2 + LDC * 4), "n"((i * 16 * LDC + j * 16) * 2 + LDC * 6)
                     : "memory");
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const bool full = (long long)(mb + 1) * BM <= a.M;
  const T* csrc = reinterpret_cast<const T*>(cw) + crow * LDC + cch * 8;
  if constexpr (EMODE == 0) {
#pragma unroll
    for (int it = 0; it < 128 / RPI; ++it) {
      const long long m = row0 + it * RPI + crow;
      Vec16<T> v;
      v.load(csrc + it * RPI * LDC);
      if (full || m < a.M) v.store(y + m * a.Cout + col0 + cch * 8);
    }
    if (a.stats) {
      __syncthreads();
      const float* red = reinterpret_cast<const float*>(smem + 8 * CWB);
      for (int o = tid; o < 2 * BN; o += 512) {
        const int which = o / BN, c = o - which * BN;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < WGM; ++r) t += red[(r * 2 + which) * BN + c];
        a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] = t;
      }
    }
  } else if constexpr (EMODE == 2) {
    // MAAI_EPI_BN_ACT (frozen statistics): out = act(r(y)*scale + shift (+ residual)) on the bf16-rounded tile — the
    // arithmetic of maai_bn_act_fwd on the stored tensor, so launch + pass and this launch give the same bits
    float q0[8], q1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col0 + cch * 8 + e;
      q0[e] = a.ep0 ? a.ep0[c] : 1.f;
      q1[e] = a.ep1 ? a.ep1[c] : 0.f;
    }
    const T* __restrict__ res = reinterpret_cast<const T*>(a.et);
    constexpr int NB = 4;
#pragma unroll
    for (int it0 = 0; it0 < 128 / RPI; it0 += NB) {
      Vec16<T> vr[NB];
      bool ok[NB];
      long long off[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const long long m = row0 + (it0 + b) * RPI + crow;
        ok[b] = full || m < a.M;
        off[b] = m * a.Cout + col0 + cch * 8;
        vr[b].zero();
        if (ok[b] && res) vr[b].load(res + off[b]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!ok[b]) continue;
        Vec16<T> v;
        v.load(csrc + (it0 + b) * RPI * LDC);
        float fv[8];
        v.get(fv);
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = fv[e] * q0[e] + q1[e];
        if (res) {
          float fr[8];
          vr[b].get(fr);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fr[e];
        }
        if (a.erelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] = fmaxf(fv[e], 0.f);
        }
        v.set(fv);
        v.store(y + off[b]);
      }
    }
  } else {
    // data-gradient epilogue (EMODE 6 of conv_igemm.h): optional accumulate, mask (1-bit array, tensor, or the unit below's
    // y*scale + shift > 0), BatchNorm-backward partial sums of the STORED (rounded) gradient
    float q0[8], q1[8], q2[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col0 + cch * 8 + e;
      q0[e] = a.ep0 ? a.ep0[c] : 0.f;
      q1[e] = a.ep1 ? a.ep1[c] : 0.f;
      q2[e] = a.ep2 ? a.ep2[c] : 0.f;
      s1[e] = 0.f;
      s2[e] = 0.f;
    }
    const T* __restrict__ et = reinterpret_cast<const T*>(a.et);
    constexpr int NB = 4;
#pragma unroll
    for (int it0 = 0; it0 < 128 / RPI; it0 += NB) {
      Vec16<T> vo[NB], vy[NB], vm[NB];
      unsigned mb8[NB];
      bool ok[NB];
      long long off[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const long long m = row0 + (it0 + b) * RPI + crow;
        ok[b] = full || m < a.M;
        off[b] = m * a.Cout + col0 + cch * 8;
        mb8[b] = 0;
        if (ok[b]) {
          if (a.accumulate) vo[b].load(y + off[b]);
          if (et) vy[b].load(et + off[b]); else vy[b].zero();   // (null: sum of the stored gradient only, see conv_igemm.h)
          if (a.mask) {
            if (a.mask_bits) mb8[b] = reinterpret_cast<const unsigned char*>(a.mask)[off[b] >> 3];
            else vm[b].load(reinterpret_cast<const T*>(a.mask) + off[b]);
          }
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!ok[b]) continue;
        Vec16<T> v;
        v.load(csrc + (it0 + b) * RPI * LDC);
        float fv[8], fo[8], fy[8];
        v.get(fv);
        if (a.accumulate) {
          vo[b].get(fo);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fo[e];
        }
        vy[b].get(fy);
        if (a.ediag) {   // the folded data gradient's fp32 diagonal: += diag[c] * relu(r(y*scale + shift)) (conv_igemm.h)
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += a.ediag[col0 + cch * 8 + e] * fmaxf(round_as<T>(fy[e] * q1[e] + q2[e]), 0.f);
        }
        if (a.mask) {
          if (a.mask_bits) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = ((mb8[b] >> e) & 1u) ? fv[e] : 0.f;
          } else {
            float fm[8];
            vm[b].get(fm);
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = fm[e] > 0.f ? fv[e] : 0.f;
          }
        } else if (a.ep1 && a.ep2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] = (fy[e] * q1[e] + q2[e]) > 0.f ? fv[e] : 0.f;
        }
        v.set(fv);
        v.get(fv);
        if (a.accumulate && a.sum_incr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] -= fo[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += fv[e];
          s2[e] += fv[e] * (fy[e] - q0[e]);
        }
        v.store(y + off[b]);
      }
    }
    // lanes l, l + CPR, ... of a wave hold the same channels: butterfly, then the two row groups through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o);
        s2[e] += __shfl_xor(s2[e], o);
      }
    }
    float* red = reinterpret_cast<float*>(smem + 8 * CWB);   // [WGM wr][2][BN]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wr * 2 + 0) * BN + wc * WN + lane * 8 + e] = s1[e];
        red[(wr * 2 + 1) * BN + wc * WN + lane * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    for (int o = tid; o < 2 * BN; o += 512) {
      const int which = o / BN, c = o - which * BN;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < WGM; ++r) t += red[(r * 2 + which) * BN + c];
      a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] = t;
    }
  }
}

template <int WGM, int WGN, int EMODE>
static int launch_pp(ConvArgs a, hipStream_t st) {
  constexpr int BM = WGM * 128, BN = WGN * 64;
  constexpr int ring = 2 * 2 * (WGM * 64 * 128) + 2 * 2 * (WGN * 32 * 128);
  constexpr int epi = 8 * 128 * (64 + 8) * 2 + WGM * 2 * BN * 4;
  constexpr int lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "conv_pp: LDS");
  a.nMB = (int)((a.M + BM - 1) / BM);
  a.nNB = a.Cout / BN;
  static int attr_lds[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&conv_pp_kernel<WGM, WGN, EMODE>), lds, attr_lds);
  MAAI_NOTE_KERNEL(conv_pp_kernel<WGM, WGN, EMODE>);
  hipLaunchKernelGGL((conv_pp_kernel<WGM, WGN, EMODE>), dim3((unsigned)(a.nMB * a.nNB)), dim3(512), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// Shapes the kernel is built for: bf16, Cin % 64 == 0, K >= 128, <= 32 taps, Cout % 128 == 0, dense output grid, an input of
// fewer than 2^31 elements, plain store (+ statistics), the data-gradient epilogue, or (256-channel tiles) the frozen-BatchNorm epilogue.  Whether it is USED is the caller's
// shape rule (conv_fwd.hip).  Rows per tile (the statistics slab has one row per tile): 256, or 512 for Cout % 256 != 0.
int maai_conv_pp_rows(int Cout) { return Cout % 256 == 0 ? 256 : 512; }

bool maai_conv_pp_supported(const ConvArgs& a, int dtype) {
  if (dtype != MAAI_BF16 || a.Cin % 64 || a.Cout % 128 || a.KH * a.KW > 32 || a.KH * a.KW * a.Cin < 128) return false;
  if (a.ostr != 1 || a.ooh != 0 || a.oow != 0 || a.OH != a.OHg || a.OW != a.OWg) return false;
  if ((long long)a.N * a.IH * a.IW * a.Cin >= (1ll << 31)) return false;
  if (a.xs || a.xb || a.a2 || a.pre_x) return false;
  if (a.x2 && (a.emode != MAAI_EPI_DGRAD_REDUCE || a.KH * a.KW != 1 || a.stride != 1 || a.cin1 % 64 || (a.Cin - a.cin1) % 64)) return false;
  if (a.emode == MAAI_EPI_STORE) return !a.accumulate && !a.mask;
  if (a.emode == MAAI_EPI_BN_ACT) return a.Cout % 256 == 0 && !a.accumulate && !a.mask;   // (the 256 x 256 tile only)
  return a.emode == MAAI_EPI_DGRAD_REDUCE;
}

int maai_conv_pp_launch(const ConvArgs& a, hipStream_t st) {
  if (a.Cout % 256 == 0) {
    if (a.emode == MAAI_EPI_DGRAD_REDUCE) return launch_pp<2, 4, 6>(a, st);
    if (a.emode == MAAI_EPI_BN_ACT) return launch_pp<2, 4, 2>(a, st);
    return launch_pp<2, 4, 0>(a, st);
  }
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) return launch_pp<4, 2, 6>(a, st);
  return launch_pp<4, 2, 0>(a, st);
}
