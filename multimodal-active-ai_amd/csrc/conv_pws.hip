// Streaming kernel for the pointwise stride-1 layers with few input channels (Cin <= 256, bf16): the HBM-bound 1x1
// convolutions of the first bottleneck stages (resnet.py:101-109 conv1 / conv3, the projection shortcuts) and the data
// gradients with the same shape.
//
// The ring kernel of conv_igemm.h keeps at most NSTAGE-1 K-steps of the A operand in flight per workgroup — 16 KB
// against an HBM latency that wants ~40 KB per CU — and re-stages A once per column tile.  Here the A operand never
// touches LDS: each wave loads the whole K extent of ITS 32 pixel rows straight into registers in MFMA layout (every
// byte of the tile in flight at once, read from HBM exactly once per launch), and the workgroup then walks ALL column
// tiles of the output, streaming only the weights (L2-resident) through a small LDS-DMA ring.  The wave grid is 4x1, so
// a wave owns full output rows: its C tile goes through a PRIVATE LDS area to 16-byte row stores with no workgroup
// barrier, and the weight prefetch for the next column tile keeps running under the epilogue.
// Normalise-on-load (XF 1; XF 2 = the two-tensor residual join with its side outputs) is applied to the registers once
// per element (the ring kernel repeats it per column tile).
// Same MFMA sequence per output element as conv_igemm_kernel (K ascending, 32 per instruction): identical bits.
#include "conv_igemm.h"

// The wait in front of every K-step's barrier.  vmcnt(N): the step's weight stage has landed.  lgkmcnt(0): every LDS read
// this wave has issued — the fragment reads of the previous step among them — has returned, so the slot that is refilled
// right after the barrier is provably no longer being read by anyone, wherever hipcc schedules the (register-only) MFMAs of
// the unrolled K loop (the "memory" clobber keeps the ds_read instructions themselves on their side of this statement).
template <int N>
__device__ __forceinline__ void wait_vm() {
#if MAAI_EXP & 2   // A/B build without the drain (scripts/build_variant.sh nodrain "-DMAAI_EXP=2"): what the drain costs
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#endif
}

// KC = Cin; BN = column tile; DIST = weight stages in flight ahead of the one being multiplied (a stage = 32 input
// channels x BN output channels).  The ring has DIST + 2 slots: the slot refilled after the barrier of step s is the
// one read in step s-2.  That its reads are over is enforced, not left to instruction scheduling: every wave drains its
// LDS reads (lgkmcnt(0), wait_vm above) before each barrier — the K loop is fully unrolled and hipcc does sink the MFMAs
// of a step below the next barrier, which raced on a DIST + 1 ring before the drain existed; the spare slot is kept.
// EMODE 0: plain store (+ statistics slab); 1: statistics slab only (the chained launch of conv_chain.hip recomputes the tensor);
// 2: MAAI_EPI_BN_ACT — frozen statistics: out = act(r(y)*scale + shift (+ residual)) applied to the bf16-rounded tile on its
// way out (the arithmetic of maai_bn_act_fwd on the stored tensor: bit-identical to launch + pass), no raw output in HBM.
// 6: MAAI_EPI_DGRAD_REDUCE for the data gradient of an expanding pointwise layer whose consumer is FOLDED (round 4: conv1's data
// gradient flowing into the block's shortcut gradient, resnet.py:101 backwards): out = (acc (+ out)) * mask bit, one slab row per
// 128 pixels with the sum of the stored values (the first BatchNorm-backward sum; the second is not wanted: the unit below
// never reads its raw output, engine._FOLD) — RES = accumulate (the previous content is the "residual", read from the output
// tensor itself a column tile ahead), the mask bytes travel with it.
// RES (EMODE 2): a residual tensor is added; its tile is requested a whole column tile ahead (before the K loop), and the
// loads enter the vmcnt bookkeeping next to the stores they follow.  BITS (EMODE 2): the 1-bit ReLU mask of the stored output
// goes to a.x_bits (one byte per 16-byte chunk, the layout of maai_bn_act_fwd_mask) — the training forward of a unit whose
// raw output is never stored (engine._FOLD) keeps it for the backward pass.
// Workgroups per CU (the register budget): the epilogues that hold a prefetched residual / previous-content tile (EMODE 2 + RES,
// EMODE 6) need ~40 more registers than the plain store — budgets chosen so that no instantiation on the default path spills
// (checked with -Rpass-analysis=kernel-resource-usage; round 4 found 6-18 spilled registers per lane on three of them).
template <int KC, int BN, int XF, int EMODE, bool RES>
constexpr int pws_min_blocks() {
  if (BN == 128) return KC == 256 ? 2 : 3;
  if (KC == 256) return (XF == 2 || (EMODE == 6 && RES)) ? 2 : 3;
  if (KC == 128 && RES && (EMODE == 2 || EMODE == 6)) return 3;
  return 4;
}
template <int KC, int BN, int DIST, int XF, int EMODE, bool RES = false, bool BITS = false>
__global__ __launch_bounds__(256, (pws_min_blocks<KC, BN, XF, EMODE, RES>())) void conv_pws_kernel(ConvArgs a) {
  typedef bf16_t T;
  constexpr int TM = 2, BM = 64 * TM, TN = BN / 16, KT = KC / 32, BR = BN / 64, STAGE = BN * 64;
  constexpr int LDC = BN + 8;              // private C tile row pitch (elements)
  constexpr int CW = 16 * LDC * 2;         // bytes of one wave's 16-row C area
  constexpr int CPR = BN / 8;              // 16-byte chunks per output row of the column tile
  constexpr int RPI = 64 / CPR;            // rows one wave-wide 16-byte access covers
  constexpr int NIT = 16 / RPI;            // such accesses per 16-row group
  constexpr int NST = (EMODE == 1 ? 0 : TM * NIT) * ((RES ? 2 : 1) + (BITS ? 1 : 0) + (EMODE == 6 ? 1 : 0));   // global stores (+ residual / previous-content loads, + mask bytes written or read) per wave per column tile
  constexpr int NSLOT = DIST + 2;
  constexpr int RING = NSLOT * STAGE;
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + RING + 4 * CW);       // [4 waves x 4 lane groups][2][BN]
  float* xcoef = reinterpret_cast<float*>(smem + RING);                                                  // XF: xs | xt, KC floats each — in the C area, which nothing else uses before the first epilogue

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int mb = xcd_remap(blockIdx.x, a.nMB);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  const int nCT = a.Cout / BN;
  const int S = nCT * KT;                  // weight stages of this workgroup
  const bool full = (long long)(mb + 1) * BM <= a.M;

  if constexpr (XF != 0) {
    const int nco = (XF == 2 && a.xs2) ? 4 * KC : 2 * KC;
    for (int i = tid; i < nco; i += 256) {
      const int which = i / KC, c = i - which * KC;
      xcoef[i] = which == 0 ? a.xs[c] : (which == 1 ? a.xt[c] : (which == 2 ? a.xs2[c] : a.xt2[c]));
    }
  }

  // ---- the A operand: rows mb*128 + wid*32 + i*16 + (lane & 15), channels kt*32 + (lane >> 4)*8 .. +8 ----
  const long long arow0 = (long long)mb * BM + widu * (16 * TM);
  uint4 areg[TM][KT];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = arow0 + i * 16 + (lane & 15);
    const T* src = x + (m < a.M ? m : a.M - 1) * KC + (lane >> 4) * 8;   // unconditional loads; rows past the end are zeroed below
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) areg[i][kt] = ld16_nt(src + kt * 32);
  }
  if (!full) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if (arow0 + i * 16 + (lane & 15) >= a.M) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) areg[i][kt] = make_uint4(0, 0, 0, 0);
      }
  }
  if constexpr (XF != 0) {
    // normalise-on-load, once per element: act(x*xs + xt) — or, XF 2, the residual join act((x*xs + xt) + r(xb*xs2 + xt2))
    // whose result (and 1-bit ReLU mask) is also handed back for the shortcut; maai_bn_act_fwd / _fwd2 arithmetic
    uint4 breg[TM][XF == 2 ? KT : 1];
    if constexpr (XF == 2) {
      const T* __restrict__ xb = reinterpret_cast<const T*>(a.xb);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = arow0 + i * 16 + (lane & 15);
        const T* src = xb + (m < a.M ? m : a.M - 1) * KC + (lane >> 4) * 8;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) breg[i][kt] = ld16_nt(src + kt * 32);
      }
    }
    __syncthreads();  // the coefficients are in LDS (no LDS-DMA is in flight yet)
    const bool join2 = XF == 2 && a.xs2 != nullptr;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float qs[8], qt[8], qs2[8], qt2[8];
      const float* cs = xcoef + kt * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        qs[e] = cs[e];
        qt[e] = cs[KC + e];
      }
      if (join2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qs2[e] = cs[2 * KC + e];
          qt2[e] = cs[3 * KC + e];
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m = arow0 + i * 16 + (lane & 15);
        const bool ok = m < a.M;
        Vec16<T> v, w2;
        v.raw = areg[i][kt];
        if constexpr (XF == 2) {
          w2.raw = breg[i][kt];
          const bool wb = a.x_out != nullptr && a.x_bits != nullptr;
          const unsigned b = XfMath<T>::template run<true>(v, w2, qs, qt, join2 ? qs2 : nullptr, qt2, a.x_relu, wb);
          if (ok && a.x_out) {
            const long long goff = m * KC + kt * 32 + (lane >> 4) * 8;
            v.store_nt(reinterpret_cast<T*>(a.x_out) + goff);
            if (wb) a.x_bits[goff >> 3] = (unsigned char)b;
          }
        } else {
          XfMath<T>::template run<false>(v, w2, qs, qt, nullptr, nullptr, a.x_relu, false);
        }
        areg[i][kt] = ok ? v.raw : make_uint4(0, 0, 0, 0);  // rows past the end stay zero
      }
    }
  }

  // ---- EMODE 2: scale | shift of all output channels in LDS (the statistics scratch, unused here); residual prefetch ----
  float* ecoef = red;
  uint4 rv[((EMODE == 2 || EMODE == 6) && RES) ? TM * NIT : 1];
  unsigned mbit[EMODE == 6 ? TM * NIT : 1];
  auto load_res = [&](int ct) {   // the residual tile of column tile ct, in the epilogue's row-store layout
    if constexpr (EMODE == 6) {
      const unsigned char* __restrict__ mk = reinterpret_cast<const unsigned char*>(a.mask);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m0 = arow0 + i * 16 + lane / CPR;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          long long m = m0 + it * RPI;
          if (!full && m >= a.M) m = a.M - 1;   // (unconditional loads keep the vmcnt bookkeeping exact; such rows are never stored)
          const long long off = m * a.Cout + ct * BN + (lane % CPR) * 8;
          if constexpr (RES) rv[i * NIT + it] = ld16_nt(y + off);
          mbit[i * NIT + it] = mk[off >> 3];
        }
      }
    }
    if constexpr (EMODE == 2 && RES) {
      const T* __restrict__ res = reinterpret_cast<const T*>(a.et);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long long m0 = arow0 + i * 16 + lane / CPR;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const long long m = m0 + it * RPI;
          rv[i * NIT + it] = make_uint4(0, 0, 0, 0);
          if (full || m < a.M) rv[i * NIT + it] = ld16_nt(res + m * a.Cout + ct * BN + (lane % CPR) * 8);
        }
      }
    }
  };
  if constexpr (EMODE == 2) {
    for (int i = tid; i < 2 * a.Cout; i += 256) ecoef[i] = i < a.Cout ? (a.ep0 ? a.ep0[i] : 1.f) : (a.ep1 ? a.ep1[i - a.Cout] : 0.f);
    load_res(0);   // (issued before any weight stage: older than everything the K loop waits for)
  }
  if constexpr (EMODE == 6) load_res(0);

  // ---- weight stages: stage s = (column tile s / KT, K-step s % KT), 64-byte rows, swizzled like conv_igemm ----
  const int r0 = tid >> 2;
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);
  const T* wsrc = w + (long long)r0 * KC + chunk * 8;   // next stage to issue: + (ict*BN + 64 i) * KC + ikt*32
  int ikt = 0, islot = 0, issued = 0;
  auto issue_b = [&]() {
    char* sb = smem + islot * STAGE + widu * 1024;
#pragma unroll
    for (int i = 0; i < BR; ++i) dma16<true>(wsrc + (long long)(64 * i) * KC, sb + i * 4096);
    ++issued;
    if (++islot == NSLOT) islot = 0;
    if (++ikt == KT) {
      ikt = 0;
      wsrc += (long long)BN * KC - (KT - 1) * 32;
    } else {
      wsrc += 32;
    }
  };
  const int pre = S < DIST ? S : DIST;
  for (int s = 0; s < pre; ++s) issue_b();

  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) << 1)) << 4);
  char* cw = smem + RING + widu * CW;      // this wave's private C area
  const uint32_t cwa = (uint32_t)(uintptr_t)(cw + (((lane >> 4) * 4) * LDC + (lane & 15)) * 2);
  float* sred = red + ((widu * 4 + (lane >> 4)) * 2) * BN + (lane & 15);

  auto finish_stats = [&](int ct) {  // after a barrier that follows the epilogue of column tile ct
    if constexpr (EMODE == 6) {
      if (tid < 2 * BN) {
        const int which = tid / BN, c = tid - which * BN;
        const float t = which ? 0.f : red[c] + red[2 * BN + c] + red[4 * BN + c] + red[6 * BN + c];   // (the second sum is not reduced here)
        a.stats[((long long)mb * 2 + which) * a.Cout + ct * BN + c] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (a.stats && EMODE != 6 && EMODE != 2) {
#pragma unroll
      for (int o = tid; o < 2 * BN; o += 256) {
        const int which = o / BN, c = o - which * BN;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[(k * 2 + which) * BN + c];
        a.stats[((long long)mb * 2 + which) * a.Cout + ct * BN + c] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  int s = 0, slot = 0;
  for (int ct = 0; ct < nCT; ++ct) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      // ---- wait until weight stage s has landed.  Issued after it (in order): the DIST-1 following stages and the
      // stores of every epilogue since — E of them, a function of (kt, ct); counting fewer than were issued only waits
      // longer, so the store term is used for full tiles only (every wave then issues exactly NST stores per epilogue)
      const int EMAX = (kt + 1 <= DIST) ? (DIST - 1 - kt) / KT + 1 : 0;   // (a constant once the loop is unrolled)
      const int rem = S - 1 - s;
      if (rem >= DIST - 1) {
        const int E = full ? (ct < EMAX ? ct : EMAX) : 0;
        if (E == 0) wait_vm<(DIST - 1) * BR>();
        else if (E == 1) wait_vm<(DIST - 1) * BR + NST>();
        else wait_vm<(DIST - 1) * BR + 2 * NST>();
      } else if (rem == 1) {
        wait_vm<BR>();
      } else {
        wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (issued < S) issue_b();   // into the slot of stage s-2
      if (kt == 0 && ct > 0) finish_stats(ct - 1);
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
      ++s;
      if (++slot == NSLOT) slot = 0;
    }

    // ---- epilogue of column tile ct: wave-private, no workgroup barrier ----
    if (a.stats && EMODE != 6 && EMODE != 2) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
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
    float dsum[EMODE == 6 ? 8 : 1];
    if constexpr (EMODE == 6) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum[e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < (EMODE == 1 ? 0 : TM); ++i) {   // (EMODE 1: statistics only, nothing is stored)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous group's reads of this area are done
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
      const long long m0 = arow0 + i * 16 + lane / CPR;
      T* dst = y + m0 * a.Cout + ct * BN + (lane % CPR) * 8;
      const T* csrc = reinterpret_cast<const T*>(cw) + (lane / CPR) * LDC + (lane % CPR) * 8;
      if constexpr (EMODE == 6) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          Vec16<T> v;
          v.load(csrc + it * RPI * LDC);
          float fv[8];
          v.get(fv);
          if constexpr (RES) {
            Vec16<T> r;
            r.raw = rv[i * NIT + it];
            float fr[8];
            r.get(fr);
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] += fr[e];
          }
          const unsigned mb8 = mbit[i * NIT + it];
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] = ((mb8 >> e) & 1u) ? fv[e] : 0.f;
          v.set(fv);
          v.get(fv);   // the rounded value being stored is what a separate reduction pass would read back
          if (full || m0 + it * RPI < a.M) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum[e] += fv[e];
            v.store(dst + (long long)it * RPI * a.Cout);
          }
        }
      } else if constexpr (EMODE == 2) {
        float q0[8], q1[8];
        {
          const float* cs = ecoef + ct * BN + (lane % CPR) * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            q0[e] = cs[e];
            q1[e] = cs[a.Cout + e];
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          Vec16<T> v;
          v.load(csrc + it * RPI * LDC);
          float fv[8];
          v.get(fv);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] = fv[e] * q0[e] + q1[e];
          if constexpr (RES) {
            Vec16<T> r;
            r.raw = rv[i * NIT + it];
            float fr[8];
            r.get(fr);
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] += fr[e];
          }
          if (a.erelu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = fmaxf(fv[e], 0.f);
          }
          v.set(fv);
          if (full || m0 + it * RPI < a.M) {
            v.store(dst + (long long)it * RPI * a.Cout);
            if constexpr (BITS) {
              const unsigned b = nonzero_bits_bf16x2(v.raw.x) | (nonzero_bits_bf16x2(v.raw.y) << 2) | (nonzero_bits_bf16x2(v.raw.z) << 4) |
                                 (nonzero_bits_bf16x2(v.raw.w) << 6);   // (post-ReLU values: non-zero == positive)
              a.x_bits[((m0 + it * RPI) * a.Cout + ct * BN + (lane % CPR) * 8) >> 3] = (unsigned char)b;
            }
          }
        }
      } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        Vec16<T> v;
        v.load(csrc + it * RPI * LDC);
        if (full || m0 + it * RPI < a.M) v.store(dst + (long long)it * RPI * a.Cout);
      }
      }
    }
    if constexpr (EMODE == 2 && RES) {
      if (ct + 1 < nCT) load_res(ct + 1);   // behind this tile's stores: one package of NST vector-memory operations per epilogue
    }
    if constexpr (EMODE == 6) {
      if (ct + 1 < nCT) load_res(ct + 1);
      // this wave's column sums over its 32 rows: lanes l, l + CPR, ... hold the same channels
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) dsum[e] += __shfl_xor(dsum[e], o);
      }
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(widu * 2) * BN + lane * 8 + e] = dsum[e];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  finish_stats(nCT - 1);
}

template <int KC, int BN, int DIST, int XF, int EMODE, bool RES = false, bool BITS = false>
static int launch_pws(ConvArgs a, hipStream_t st) {
  constexpr int lds = (DIST + 2) * BN * 64 + 4 * 16 * (BN + 8) * 2 + 32 * BN * 4;
  static_assert(4 * KC * 4 <= 4 * 16 * (BN + 8) * 2, "the coefficient table borrows the C area");
  a.nMB = (int)((a.M + 127) / 128);
  static int attr_lds[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&conv_pws_kernel<KC, BN, DIST, XF, EMODE, RES, BITS>), lds, attr_lds);
  MAAI_NOTE_KERNEL(conv_pws_kernel<KC, BN, DIST, XF, EMODE, RES, BITS>);
  hipLaunchKernelGGL((conv_pws_kernel<KC, BN, DIST, XF, EMODE, RES, BITS>), dim3((unsigned)a.nMB), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

template <int KC, int BN, int DIST>
static int pws_kd(const ConvArgs& a, hipStream_t st) {
  if (a.emode == MAAI_EPI_STATS_ONLY) {
    if (a.xb) {
      maai_set_error("conv2d_igemm: the statistics-only streaming launch takes a plain or normalised-on-load input");
      return MAAI_ERR_UNSUPPORTED;
    }
    return a.xs ? launch_pws<KC, BN, DIST, 1, 1>(a, st) : launch_pws<KC, BN, DIST, 0, 1>(a, st);
  }
  if (a.emode == MAAI_EPI_BN_ACT) {
    if (a.xb) {
      maai_set_error("conv2d_igemm: the BatchNorm epilogue of the streaming kernel takes a plain or normalised-on-load input");
      return MAAI_ERR_UNSUPPORTED;
    }
    if (a.Cout > 16 * BN) {
      maai_set_error("conv2d_igemm: the streaming kernel's BatchNorm epilogue keeps scale | shift in LDS: Cout <= 16 column tiles");
      return MAAI_ERR_UNSUPPORTED;
    }
    ConvArgs b = a;
    b.stats = nullptr;   // (the statistics scratch holds the coefficient table)
    if (a.x_bits) {      // the training forward of a folded unit: a ReLU output with its 1-bit mask (tensor input)
      if (!a.erelu || a.xs) {
        maai_set_error("conv2d_igemm: the mask output of the BatchNorm epilogue belongs to a ReLU unit with a tensor input");
        return MAAI_ERR_UNSUPPORTED;
      }
      return a.et ? launch_pws<KC, BN, DIST, 0, 2, true, true>(b, st) : launch_pws<KC, BN, DIST, 0, 2, false, true>(b, st);
    }
    if (a.et) return a.xs ? launch_pws<KC, BN, DIST, 1, 2, true>(b, st) : launch_pws<KC, BN, DIST, 0, 2, true>(b, st);
    return a.xs ? launch_pws<KC, BN, DIST, 1, 2, false>(b, st) : launch_pws<KC, BN, DIST, 0, 2, false>(b, st);
  }
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) {
    if (a.xs || a.xb || a.et || !a.mask || !a.mask_bits || !a.stats) {
      maai_set_error("conv2d_igemm: the streaming kernel's data-gradient epilogue is the sum-only form with a 1-bit mask");
      return MAAI_ERR_UNSUPPORTED;
    }
    return a.accumulate ? launch_pws<KC, BN, DIST, 0, 6, true>(a, st) : launch_pws<KC, BN, DIST, 0, 6, false>(a, st);
  }
  if (a.xb) return launch_pws<KC, BN, DIST, 2, 0>(a, st);
  if (a.xs) return launch_pws<KC, BN, DIST, 1, 0>(a, st);
  return launch_pws<KC, BN, DIST, 0, 0>(a, st);
}

// DIST = weight stages in flight ahead of the one being multiplied.  MAAI_PWS_DIST = 0 (default: 3 for 256 input channels, else 2)
// | 1 (one more: 4 / 3) — A/B knob.
template <int KC, int BN>
static int pws_k(const ConvArgs& a, hipStream_t st) {
  static const int more = getenv("MAAI_PWS_DIST") ? atoi(getenv("MAAI_PWS_DIST")) : 0;
  if (more == 1) return pws_kd<KC, BN, (KC == 256 ? 4 : 3)>(a, st);
  return pws_kd<KC, BN, (KC == 256 ? 3 : 2)>(a, st);
}

// One 128-row tile per workgroup (the statistics slab's rows).  Column tile: 64 output channels (4 workgroups per CU; 3 for Cin 256) except for Cin >= 256
// with many output channels, where the MFMA share is large enough for the 128-wide tile's fewer barriers to win
// (256->1024: 0.64 vs 0.67 ms).  MAAI_PWS_BN = 64 | 128 overrides.
int maai_conv_pws_launch(const ConvArgs& a, hipStream_t st) {
  const char* e = getenv("MAAI_PWS_BN");   // experiment knob
  const int forced = e ? atoi(e) : 0;
  // (the data-gradient epilogue with accumulate holds a previous-content tile next to the accumulators: at 128 columns and 256
  //  input channels it spills 12 registers per lane, at 64 columns none)
  const bool n128 = a.Cout % 128 == 0 && forced != 64 &&
                    (forced == 128 || (a.Cin >= 256 && a.Cout >= 512 && !(a.emode == MAAI_EPI_DGRAD_REDUCE && a.accumulate)));
  switch (a.Cin) {
    case 64: return n128 ? pws_k<64, 128>(a, st) : pws_k<64, 64>(a, st);
    case 128: return n128 ? pws_k<128, 128>(a, st) : pws_k<128, 64>(a, st);
    case 256: return n128 ? pws_k<256, 128>(a, st) : pws_k<256, 64>(a, st);
    default: break;
  }
  maai_set_error("conv2d_igemm: no streaming pointwise kernel for this channel count");
  return MAAI_ERR_UNSUPPORTED;
}
