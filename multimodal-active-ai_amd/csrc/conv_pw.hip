// Pointwise (1x1, stride 1, dense output) convolution with a DIRECT epilogue — the kernel behind every
// conv1 / conv3 / downsample-at-stride-1 of the bottlenecks (resnet.py:105,109), their data gradients, and the
// fused conv+BatchNorm units.
//
// Why a second kernel: these layers are HBM-bound, and in conv_fwd.hip their epilogue (accumulators -> bf16 ->
// LDS -> 16-byte rows -> math -> store) is VALU-bound at ~3 TB/s (measured by ablation: 600 vector instructions
// per wave against 32 MFMAs).  Here the MFMA operands are swapped — A = weights, B = pixels — so the
// accumulator tile is C[cout][pixel]: every lane owns 4 consecutive output channels of ONE pixel per 16x16 tile.
// The weight rows are staged in a permuted order (free: the LDS-DMA source address is per lane) such that two
// neighbouring tiles give the lane 8 CONSECUTIVE channels = one 16-byte NHWC access.  Residual, dz, mask and
// the output are read / written straight from / to global memory at 16 bytes per lane (64 contiguous bytes per
// pixel per instruction, a full 128-byte line per pixel per wave), the BN maths runs on the fp32 accumulators
// in registers with per-lane channel constants, and no C tile ever goes through LDS.
//
// Epilogue modes (maai_conv_epilogue): 0 store, 1 statistics only, 2 BN-apply(+residual)(+ReLU),
// 3 BN-backward reduce, 4 BN-backward apply, 5 store with accumulate and/or ReLU mask.
// The main loop is the LDS-DMA ring of conv_fwd.hip (3 slots, counted vmcnt, one raw barrier per K-step).
#include "common.h"
#include "maai_internal.h"
#include "conv_pw.h"

__device__ uint4 g_pwzero64[4];

template <int BN, int NSTAGE, int EMODE>
__global__ __launch_bounds__(256) void pw_conv_kernel(PwArgs a) {
  constexpr int BM = 128, BK = 32, EPC = 8;
  constexpr int WC = BN / 2;        // output channels per wave
  constexpr int TP = 4;             // 16-pixel tiles per wave (64 pixels)
  constexpr int TC = WC / 16;       // 16-channel tiles per wave
  constexpr int NT = TC / 2;        // tile pairs = 16-byte channel groups per lane
  constexpr int AR = BM / 64, BR = BN / 64;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int NL = AR + BR;
  constexpr bool FUSED = EMODE >= 2 && EMODE <= 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  const int wp = wid >> 1, wc = wid & 1;  // pixel half, channel half
  const int logical = xcd_remap(blockIdx.x, a.nMB * a.nNB);
  const int mb = logical / a.nNB, nb = logical - mb * a.nNB;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
  const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
  const int K = a.Cin;
  const int KT = K / BK;
  const int r0 = tid >> 2;
  const int chunk = (tid & 3) ^ (((r0 >> 3) & 1) << 1);  // inverse swizzle on the DMA source

  // ---- DMA sources (fixed per thread) ----
  const bf16_t* xp[AR];
  bool xok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const long long m = (long long)mb * BM + r0 + 64 * i;
    xok[i] = m < a.M;
    xp[i] = x + (xok[i] ? m : 0) * K + chunk * EPC;
  }
  const bf16_t* wpt[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    // LDS row R = wcR*WC + jw*16 + rho  holds output channel  wcR*WC + (jw>>1)*32 + 8*(rho>>2) + 4*(jw&1) + (rho&3)
    const int R = r0 + 64 * i;
    const int wcR = R / WC, rr = R - wcR * WC, jw = rr >> 4, rho = rr & 15;
    const int co = wcR * WC + (jw >> 1) * 32 + 8 * (rho >> 2) + 4 * (jw & 1) + (rho & 3);
    wpt[i] = w + (long long)(nb * BN + co) * K + chunk * EPC;
  }
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_pwzero64);

  auto issue_stage = [&](int kt, int slot) {
    char* sa = smem + slot * STAGE + widu * 1024;
    char* sb = sa + BM * 64;
#pragma unroll
    for (int i = 0; i < AR; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xok[i] ? xp[i] + kt * BK : zsrc),
                                       (__attribute__((address_space(3))) void*)(sa + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BR; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wpt[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sb + i * 4096), 16, 0, 0);
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int j = 0; j < TC; ++j)
#pragma unroll
    for (int i = 0; i < TP; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int foff = frow * 64 + (((lane >> 4) ^ (((frow >> 3) & 1) << 1)) << 4);
  const int pre = KT < NSTAGE - 1 ? KT : NSTAGE - 1;
  for (int s = 0; s < pre; ++s) issue_stage(s, s);
  for (int kt = 0; kt < KT; ++kt) {
    const int ahead = KT - 1 - kt;
    if (NSTAGE >= 3 && ahead >= 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < KT) issue_stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    const int slot = kt % NSTAGE;
    const char* sx = smem + slot * STAGE + (wp * 64) * 64 + foff;
    const char* sw = smem + slot * STAGE + BM * 64 + (wc * WC) * 64 + foff;
    bf16x8 xf[TP], wf[TC];
#pragma unroll
    for (int i = 0; i < TP; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(sx + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < TC; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sw + j * 16 * 64);
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int i = 0; i < TP; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
  }

  // ---- direct epilogue: lane (g, q) owns pixel q of every pixel tile and channels t*32 + 8g .. +7 ----
  const int g = lane >> 4, q = lane & 15;
  const int cl0 = wc * WC + 8 * g;                       // + t*32: channel inside the BN-wide tile
  const long long mrow0 = (long long)mb * BM + wp * 64 + q;  // + i*16
  bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(a.y);

  float p0[NT][8], p1[NT][8], p2[NT][8];
  if constexpr (FUSED) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = nb * BN + cl0 + t * 32;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p0[t][e] = a.ep0 ? a.ep0[c + e] : (EMODE == 2 ? 1.f : 0.f);
        p1[t][e] = ((EMODE == 2 || EMODE == 4) && a.ep1) ? a.ep1[c + e] : 0.f;
        p2[t][e] = (EMODE == 4 && a.ep2) ? a.ep2[c + e] : 0.f;
      }
    }
  }
  constexpr bool REDUCE = (EMODE == 0 || EMODE == 1 || EMODE == 2 || EMODE == 3);  // may produce per-channel partial sums
  const bool do_red = REDUCE && a.stats != nullptr;
  float s1[NT][8], s2[NT][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[t][e] = 0.f; s2[t][e] = 0.f; }

#pragma unroll
  for (int i = 0; i < TP; ++i) {
    const long long m = mrow0 + i * 16;
    const bool rowok = m < a.M;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[2 * t][i][e];
        v[4 + e] = acc[2 * t + 1][i][e];
      }
      if constexpr (EMODE <= 2) {
        if (do_red) {  // BatchNorm statistics of the fp32 results (rows past M are exact zeros)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[t][e] += v[e];
            s2[t][e] = __builtin_fmaf(v[e], v[e], s2[t][e]);
          }
        }
      }
      if constexpr (EMODE == 1) continue;
      const long long off = m * a.Cout + nb * BN + cl0 + t * 32;
      // y as the unfused path would have stored it: rounded to bf16
      Vec16<bf16_t> yv;
      yv.set(v);
      if constexpr (EMODE == 0) {
        if (rowok) yv.store(y + off);
        continue;
      }
      float yr[8];
      yv.get(yr);
      if constexpr (EMODE == 5) {
        if (!rowok) continue;
        if (a.accumulate) {
          Vec16<bf16_t> o;
          o.load(y + off);
          float fo[8];
          o.get(fo);
#pragma unroll
          for (int e = 0; e < 8; ++e) yr[e] += fo[e];
        }
        if (a.mask) {
          Vec16<bf16_t> mk;
          mk.load(reinterpret_cast<const bf16_t*>(a.mask) + off);
          float fm[8];
          mk.get(fm);
#pragma unroll
          for (int e = 0; e < 8; ++e) yr[e] = fm[e] > 0.f ? yr[e] : 0.f;
        }
        Vec16<bf16_t> ov;
        ov.set(yr);
        ov.store(y + off);
      } else if constexpr (EMODE == 2) {
        if (!rowok) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) yr[e] = yr[e] * p0[t][e] + p1[t][e];
        if (a.et) {
          Vec16<bf16_t> r;
          r.load(reinterpret_cast<const bf16_t*>(a.et) + off);
          float fr[8];
          r.get(fr);
#pragma unroll
          for (int e = 0; e < 8; ++e) yr[e] += fr[e];
        }
        if (a.erelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) yr[e] = fmaxf(yr[e], 0.f);
        }
        Vec16<bf16_t> ov;
        ov.set(yr);
        ov.store(y + off);
      } else if constexpr (EMODE == 3 || EMODE == 4) {
        if (!rowok) continue;
        Vec16<bf16_t> dzv;
        dzv.load(reinterpret_cast<const bf16_t*>(a.et) + off);
        float dz[8];
        dzv.get(dz);
        if constexpr (EMODE == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[t][e] += dz[e];
            s2[t][e] += dz[e] * (yr[e] - p0[t][e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) yr[e] = p0[t][e] * dz[e] - p1[t][e] - p2[t][e] * yr[e];
          Vec16<bf16_t> ov;
          ov.set(yr);
          ov.store(y + off);
        }
      }
    }
  }

  if constexpr (REDUCE) {
    if (do_red || EMODE == 3) {
      // per-channel sums over the tile's 128 pixels: transpose through LDS (the ring is free now), then
      // 2*BN threads add 32 partials (2 pixel halves x 16 lanes) each.  red[which][channel][slot]
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);
      const int slot = wp * 16 + q;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = cl0 + t * 32 + e;
          red[(c)*32 + slot] = s1[t][e];
          red[(BN + c) * 32 + slot] = s2[t][e];
        }
      __syncthreads();
      if (tid < 2 * BN) {
        const float4* r4 = reinterpret_cast<const float4*>(red + tid * 32);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 f = r4[k];
          s += (f.x + f.y) + (f.z + f.w);
        }
        const int which = tid / BN, c = tid - which * BN;
        a.stats[((long long)mb * 2 + which) * a.Cout + nb * BN + c] = s;
      }
    }
  }
}

template <int BN, int NSTAGE, int EMODE>
static int launch_pw_e(const PwArgs& a, hipStream_t st) {
  constexpr int ring = NSTAGE * (128 + BN) * 64;
  constexpr int redb = 2 * BN * 32 * 4;
  constexpr int lds = ring > redb ? ring : redb;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_conv_kernel<BN, NSTAGE, EMODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((pw_conv_kernel<BN, NSTAGE, EMODE>), dim3((unsigned)((long long)a.nMB * a.nNB)), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

template <int BN, int NSTAGE>
static int launch_pw_n(const PwArgs& a, int emode, hipStream_t st) {
  switch (emode) {
    case 0: return launch_pw_e<BN, NSTAGE, 0>(a, st);
    case 1: return launch_pw_e<BN, NSTAGE, 1>(a, st);
    case 2: return launch_pw_e<BN, NSTAGE, 2>(a, st);
    case 3: return launch_pw_e<BN, NSTAGE, 3>(a, st);
    case 4: return launch_pw_e<BN, NSTAGE, 4>(a, st);
    default: return launch_pw_e<BN, NSTAGE, 5>(a, st);
  }
}

// emode: 0..4 as in maai_conv_epilogue; plain stores with accumulate / mask are routed to 5 here
int maai_pw_conv_launch(PwArgs a, int emode, hipStream_t st) {
  if (emode == 0 && (a.accumulate || a.mask)) emode = 5;
  a.nMB = (int)((a.M + 127) / 128);
  const int kt = a.Cin / 32;
  if (a.Cout % 128 == 0) {
    a.nNB = a.Cout / 128;
    return kt <= 2 ? launch_pw_n<128, 2>(a, emode, st) : launch_pw_n<128, 3>(a, emode, st);
  }
  a.nNB = a.Cout / 64;
  return kt <= 2 ? launch_pw_n<64, 2>(a, emode, st) : launch_pw_n<64, 3>(a, emode, st);
}
