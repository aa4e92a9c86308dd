// Block-boundary chain kernel (bf16, layer 1 of the bottleneck ResNets): the last convolution of a block, its
// BatchNorm, the residual join and the FIRST convolution of the next block in one launch —
//     y3 = conv3(a2)                       (resnet.py:118, 64 -> 256 channels, recomputed here, never read from HBM)
//     out = relu(bn3(y3) + shortcut)       (resnet.py:119-133; handed back once: the next block's identity shortcut and
//                                           the backward pass read it) + its 1-bit ReLU mask
//     y1' = conv1'(out)                    (resnet.py:101 of the next block, 256 -> 64 / 128) + BatchNorm partial sums
// bn3's batch statistics must exist before anything can be normalised, so conv3 has run once before as a statistics-only
// launch of the streaming kernel (conv_pws.hip, EMODE 1: reads a2, stores nothing).  What this saves per block and
// forward pass: writing y3 (6.6 GB at 224^2 x 256 images) and reading it back — the raw tensor is stored here only when
// the backward pass will need it (pre_y_out).
//
// Layout as in conv_pws.hip: a workgroup owns 128 pixel rows, a wave 32 of them; a2 AND the shortcut are loaded into
// registers at the start, the shortcut straight into the registers that will hold the joined activation in MFMA A layout
// (16 + 64 KB in flight per workgroup).  GEMM 1 walks the four 64-channel column tiles of y3; after each, the tile goes
// through the wave's private LDS area and comes back as the 16-byte chunks (row = lane & 15, chunk = lane >> 4 (+4)) that
// ARE the A fragments of GEMM 2, where bn3 + shortcut + ReLU are applied in place (arithmetic and roundings of
// maai_bn_act_fwd / _fwd2 on the bf16-rounded y3: bit-identical to the unchained launches).  Weights of both GEMMs stream
// through one LDS-DMA ring of DIST + 2 slots.  vmcnt bookkeeping is dynamic: a uniform counter of vector-memory
// instructions issued, the value it had after each stage's issue, and a switch onto the immediate forms.
#include "conv_igemm.h"
#include <stdlib.h>

__device__ __forceinline__ void chain_wait_vm(int n) {   // wait until at most n vector-memory operations are outstanding
  switch (n < 0 ? 0 : (n > 23 ? 23 : n)) {                // (waiting for fewer than allowed is always safe)
  // (lgkmcnt(0): this wave's fragment reads of the previous step have returned before it enters the barrier that precedes
  //  the refill of their slot — the ring invariant is enforced here, not left to where hipcc puts the MFMAs)
#if MAAI_EXP & 2   // A/B build without the drain
#define MAAI_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#else
#define MAAI_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory"); break;
#endif
    MAAI_WVM(0) MAAI_WVM(1) MAAI_WVM(2) MAAI_WVM(3) MAAI_WVM(4) MAAI_WVM(5) MAAI_WVM(6) MAAI_WVM(7) MAAI_WVM(8) MAAI_WVM(9)
    MAAI_WVM(10) MAAI_WVM(11) MAAI_WVM(12) MAAI_WVM(13) MAAI_WVM(14) MAAI_WVM(15) MAAI_WVM(16) MAAI_WVM(17) MAAI_WVM(18)
    MAAI_WVM(19) MAAI_WVM(20) MAAI_WVM(21) MAAI_WVM(22) MAAI_WVM(23)
#undef MAAI_WVM
  }
}

// KC1 = channels of a2 (conv3's input); K2 = 4*KC1 = channels of the block output = conv1' input.
// PROJ: the shortcut is a raw projection branch normalised on load (xs2/xt2); BITS: store the 1-bit mask; KEEPY: store y3.
template <int KC1, int DIST, bool PROJ, bool BITS, bool KEEPY>
__global__ __launch_bounds__(256, 3) void conv_chain_kernel(ConvArgs a) {
  typedef bf16_t T;
  constexpr int TM = 2, BN = 64, TN = 4, K2 = 4 * KC1, KT1 = KC1 / 32, KT2 = K2 / 32, NCT1 = K2 / BN, STAGE = BN * 64;
  constexpr int NSLOT = DIST + 2, RING = NSLOT * STAGE;
  constexpr int LDC = BN + 8, CW = 16 * LDC * 2;         // wave-private C area: 16 rows
  constexpr int S1 = NCT1 * KT1;                         // weight stages of GEMM 1
  constexpr int NST1 = 2 * (1 + (BITS ? 1 : 0) + (KEEPY ? 1 : 0));  // stores per 16-row group of a GEMM-1 epilogue
  constexpr int NST2 = 2;                                // ... of a GEMM-2 epilogue (64 columns: 8 rows per instruction)
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem + RING + 4 * CW);   // [4 waves x 4 lane groups][2][BN]
  float* coef = red + 32 * BN;   // pre_xs | pre_xt (KC1 each) | xs | xt (| xs2 | xt2) (K2 each)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int mb = xcd_remap(blockIdx.x, a.nMB);
  const T* __restrict__ px = reinterpret_cast<const T*>(a.pre_x);
  const T* __restrict__ w3 = reinterpret_cast<const T*>(a.pre_w);
  const T* __restrict__ w1 = reinterpret_cast<const T*>(a.w);
  const T* __restrict__ sc = reinterpret_cast<const T*>(a.xb);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  T* __restrict__ jout = reinterpret_cast<T*>(a.x_out);
  T* __restrict__ y3out = reinterpret_cast<T*>(a.pre_y_out);
  const int nCT2 = a.Cout / BN;
  const int S = S1 + nCT2 * KT2;
  const bool full = (long long)(mb + 1) * 128 <= a.M;
  const bool prexf = a.pre_xs != nullptr;

  // ---- operands into registers: a2 (A of GEMM 1), then the coefficient tables, then the shortcut ----
  const long long arow0 = (long long)mb * 128 + widu * 32;
  const int g = lane >> 4, li = lane & 15;
  uint4 areg[TM][KT1], jreg[TM][KT2];
  long long mrow[TM];
  bool rok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = arow0 + i * 16 + li;
    rok[i] = m < a.M;
    mrow[i] = rok[i] ? m : a.M - 1;
    const T* src = px + mrow[i] * KC1 + g * 8;
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) areg[i][kt] = ld16_nt(src + kt * 32);
  }
  {
    const int nco = 2 * KC1 + (PROJ ? 4 : 2) * K2;
    for (int i = tid; i < nco; i += 256) {
      float v;
      if (i < 2 * KC1) {
        v = prexf ? (i < KC1 ? a.pre_xs[i] : a.pre_xt[i - KC1]) : 0.f;
      } else {
        const int j = i - 2 * KC1, which = j / K2, c = j - which * K2;
        v = which == 0 ? a.xs[c] : (which == 1 ? a.xt[c] : (which == 2 ? a.xs2[c] : a.xt2[c]));
      }
      coef[i] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const T* src = sc + mrow[i] * K2 + g * 8;
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) jreg[i][kt] = ld16_nt(src + kt * 32);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // the coefficient tables are in LDS (the shortcut loads stay in flight)
  if (prexf) {
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) {
      float qs[8], qt[8];
      const float* cs = coef + kt * 32 + g * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        qs[e] = cs[e];
        qt[e] = cs[KC1 + e];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        Vec16<T> v, w2;
        v.raw = areg[i][kt];
        XfMath<T>::template run<false>(v, w2, qs, qt, nullptr, nullptr, a.pre_relu, false);
        areg[i][kt] = v.raw;
      }
    }
  }
  if (!full) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if (!rok[i]) {
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) areg[i][kt] = make_uint4(0, 0, 0, 0);
      }
  }

  // ---- weight stages: s < S1: conv3 (column tile s / KT1, K-step s % KT1); then conv1' ----
  const int r0 = tid >> 2;
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);
  const T* wsrc = w3 + (long long)r0 * KC1 + chunk * 8;
  int ikt = 0, islot = 0, issued = 0;
  int vm_issued = 0;             // vector-memory instructions this wave has issued that the waits below account for
  int mark[DIST];                // vm_issued right after the issue of each stage still to be multiplied, oldest first
  auto issue_b = [&](int qpos) { // (qpos: where the stage goes in that queue — a constant at every call site)
    dma16<true>(wsrc, smem + islot * STAGE + widu * 1024);
    vm_issued += 1;
#pragma unroll
    for (int k = 0; k < DIST; ++k)
      if (k == qpos) mark[k] = vm_issued;
    ++issued;
    if (++islot == NSLOT) islot = 0;
    ++ikt;
    if (issued < S1) {
      if (ikt == KT1) { ikt = 0; wsrc += (long long)BN * KC1 - (KT1 - 1) * 32; } else { wsrc += 32; }
    } else if (issued == S1) {
      ikt = 0;
      wsrc = w1 + (long long)r0 * K2 + chunk * 8;
    } else {
      if (ikt == KT2) { ikt = 0; wsrc += (long long)BN * K2 - (KT2 - 1) * 32; } else { wsrc += 32; }
    }
  };
#pragma unroll
  for (int s = 0; s < DIST; ++s) issue_b(s);   // (S >= S1 + KT2 > DIST)

  const int foff = li * 64 + ((g ^ (((li >> 3) & 1) << 1)) << 4);
  char* cw = smem + RING + widu * CW;
  const uint32_t cwa = (uint32_t)(uintptr_t)(cw + ((g * 4) * LDC + li) * 2);
  float* sred = red + ((widu * 4 + g) * 2) * BN + li;
  int slot = 0;
  // one K-step: wait for its stage, barrier, refill the slot read two steps ago, read the B fragments
  auto step_open = [&](frag_t (&bfr)[TN]) {
    chain_wait_vm(vm_issued - mark[0]);
#pragma unroll
    for (int k = 0; k + 1 < DIST; ++k) mark[k] = mark[k + 1];
    __builtin_amdgcn_s_barrier();
    if (issued < S) issue_b(DIST - 1);   // (while stages remain, DIST - 1 are outstanding here)
    const char* sb = smem + slot * STAGE + foff;
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(sb + j * 16 * 64);
    if (++slot == NSLOT) slot = 0;
  };
  auto c_write = [&](const f32x4 (&acc)[TN]) {   // one 16-row group of accumulators -> the private C area (bf16)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 v = acc[j];
      const uint32_t p01 = pack_bf16x2(v[0], v[1]);
      const uint32_t p23 = pack_bf16x2(v[2], v[3]);
      asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p01), "n"(j * 32),
                   "n"(j * 32 + LDC * 2)
                   : "memory");
      asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(cwa), "v"(p23),
                   "n"(j * 32 + LDC * 4), "n"(j * 32 + LDC * 6)
                   : "memory");
    }
  };

  // ================= GEMM 1: y3 = a2 . W3^T, joined in place into jreg =================
#pragma unroll
  for (int ct = 0; ct < NCT1; ++ct) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) {
      frag_t bfr[TN];
      step_open(bfr);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const frag_t af = __builtin_bit_cast(frag_t, areg[i][kt]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af, bfr[j], acc[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      c_write(acc[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ch = ct * BN + (g + 4 * q) * 8;   // first of this lane's 8 channels
        Vec16<T> v, w2;
        v.load(reinterpret_cast<const T*>(cw) + li * LDC + (g + 4 * q) * 8);
        const long long goff = mrow[i] * K2 + ch;
        if constexpr (KEEPY) {
          if (full || rok[i]) v.store_nt(y3out + goff);
        }
        float qs[8], qt[8], qs2[8], qt2[8];
        const float* cs = coef + 2 * KC1 + ch;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qs[e] = cs[e];
          qt[e] = cs[K2 + e];
        }
        if constexpr (PROJ) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            qs2[e] = cs[2 * K2 + e];
            qt2[e] = cs[3 * K2 + e];
          }
        }
        w2.raw = jreg[i][ct * 2 + q];
        const unsigned b = XfMath<T>::template run<true>(v, w2, qs, qt, PROJ ? qs2 : nullptr, qt2, a.x_relu, BITS);
        if (full || rok[i]) {
          v.store_nt(jout + goff);
          if constexpr (BITS) a.x_bits[goff >> 3] = (unsigned char)b;
        }
        jreg[i][ct * 2 + q] = (full || rok[i]) ? v.raw : make_uint4(0, 0, 0, 0);   // rows past the end stay zero
      }
      if (full) vm_issued += NST1;
    }
  }

  // ================= GEMM 2: y1' = out . W1^T (+ BatchNorm partial sums) =================
  auto finish_stats = [&](int ct) {
    if (a.stats) {
      if (tid < 2 * BN) {
        const int which = tid / BN, c = tid - which * BN;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[(k * 2 + which) * BN + c];
        a.stats[((long long)mb * 2 + which) * a.Cout + ct * BN + c] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  for (int ct = 0; ct < nCT2; ++ct) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt) {
      frag_t bfr[TN];
      step_open(bfr);
      if (kt == 1 && ct > 0) finish_stats(ct - 1);   // (a barrier after every wave's partials, a barrier before the next ones)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const frag_t af = __builtin_bit_cast(frag_t, jreg[i][kt]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(af, bfr[j], acc[i][j]);
      }
    }
    if (a.stats) {
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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      c_write(acc[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long m0 = arow0 + i * 16 + (lane >> 3);
      T* dst = y + m0 * a.Cout + ct * BN + (lane & 7) * 8;
      const T* csrc = reinterpret_cast<const T*>(cw) + (lane >> 3) * LDC + (lane & 7) * 8;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        Vec16<T> v;
        v.load(csrc + it * 8 * LDC);
        if (full || m0 + it * 8 < a.M) v.store(dst + (long long)it * 8 * a.Cout);
      }
      if (full) vm_issued += NST2;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  finish_stats(nCT2 - 1);
}

template <int KC1, bool PROJ, bool BITS, bool KEEPY, int DIST>
static int launch_chain_d(ConvArgs a, hipStream_t st) {
  constexpr int lds = (DIST + 2) * 64 * 64 + 4 * 16 * 72 * 2 + 32 * 64 * 4 + (2 * KC1 + 16 * KC1) * 4;
  a.nMB = (int)((a.M + 127) / 128);
  static int attr_lds[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&conv_chain_kernel<KC1, DIST, PROJ, BITS, KEEPY>), lds, attr_lds);
  MAAI_NOTE_KERNEL(conv_chain_kernel<KC1, DIST, PROJ, BITS, KEEPY>);
  hipLaunchKernelGGL((conv_chain_kernel<KC1, DIST, PROJ, BITS, KEEPY>), dim3((unsigned)a.nMB), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// DIST = weight stages in flight ahead of the one being multiplied (ring of DIST + 2 slots of 4 KB).  MAAI_CHAIN_DIST = 3 | 5 (A/B knob).
template <int KC1, bool PROJ, bool BITS, bool KEEPY>
static int launch_chain(ConvArgs a, hipStream_t st) {
  static const int dist = getenv("MAAI_CHAIN_DIST") ? atoi(getenv("MAAI_CHAIN_DIST")) : 3;
  if (dist == 5) return launch_chain_d<KC1, PROJ, BITS, KEEPY, 5>(a, st);
  return launch_chain_d<KC1, PROJ, BITS, KEEPY, 3>(a, st);
}

template <int KC1, bool PROJ>
static int chain_p(const ConvArgs& a, hipStream_t st) {
  const bool bits = a.x_bits != nullptr, keepy = a.pre_y_out != nullptr;
  if (bits) return keepy ? launch_chain<KC1, PROJ, true, true>(a, st) : launch_chain<KC1, PROJ, true, false>(a, st);
  return keepy ? launch_chain<KC1, PROJ, false, true>(a, st) : launch_chain<KC1, PROJ, false, false>(a, st);
}

// a.pre_x / a.pre_w: conv3's input [M][pre_cin] and weights [Cin][pre_cin]; a.Cin = 4 * pre_cin; a.xs/xt: bn3; a.xb:
// shortcut; a.x_out: joined activation (required); one statistics-slab row per 128 pixels
int maai_conv_chain_launch(const ConvArgs& a, hipStream_t st) {
  if (a.pre_cin == 64 && a.Cin == 256) return a.xs2 ? chain_p<64, true>(a, st) : chain_p<64, false>(a, st);
  maai_set_error("conv2d_igemm: the chained launch is built for 64 -> 256 -> Cout");
  return MAAI_ERR_UNSUPPORTED;
}
