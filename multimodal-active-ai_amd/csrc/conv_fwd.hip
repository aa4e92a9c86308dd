// Implicit-GEMM convolution on MFMA for gfx950 (K2/K3/K4 of SURVEY §2.3; also
// the data-gradient and the MLP GEMMs, which are the same contraction).
//
//   y[m][co] (+)= sum_{kh,kw,ci} x[n, oh*s-ph+kh, ow*s-pw+kw, ci] * w[co][kh][kw][ci]
//
// GEMM view: M = N*OHg*OWg pixels, N = Cout, K = KH*KW*Cin.  NHWC activations,
// [Cout][KH][KW][Cin] weights (K-contiguous per output channel), so every MFMA
// A/B fragment is one 16-byte run in memory.
//
// Tile: BM x BN per 256-thread workgroup (4 waves as 2x2), K-step = 64 bytes per
// row (32 bf16 / 16 f32).  Staging is LDS-DMA (global_load_lds_dwordx4: no
// staging VGPRs) into a 4-slot ring with THREE K-steps in flight behind a
// counted s_waitcnt vmcnt(N) and ONE raw s_barrier per K-step
// (cdna_hip_programming.md §5 "Pipelining across barriers", T3/T4).  The DMA
// destination is lane-linear, so the st_16x32 XOR swizzle that makes the
// ds_read_b128 fragment reads bank-conflict free (T2) is applied to the per-lane
// SOURCE address and to the read (rule 21).  Padding / out-of-range rows source
// a 64-byte zero page instead of being predicated.
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16 storage) or 4x v_mfma_f32_16x16x4_f32
// (f32 storage, exact fp32 — the parity mode), fp32 accumulation in both.
// Epilogue: accumulators -> LDS -> 16-byte row-contiguous stores (full 128-B
// lines), optional read-modify-write accumulate and strided scatter (used by
// the stride-2 data gradients), optional per-channel sum / sum-of-squares
// partials for BatchNorm (one deterministic slab row per M-block, no atomics).
#include "conv_igemm.h"
#include <string.h>

template <typename T, int BM, int BN, int NSTAGE, int EMODE>
static int launch_conv_e(const ConvArgs& a, hipStream_t st) {
  const bool pw = (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_h == 0 && a.pad_w == 0 && a.OHg == a.IH && a.OWg == a.IW);
  if constexpr (EMODE == 1 || EMODE == 3 || EMODE == 4) {
    // the recompute epilogues (statistics only, BN-backward reduce / apply) exist for pointwise layers; BN_ACT (2) is
    // general: with frozen statistics any convolution can normalise, add the shortcut and activate in its epilogue
    if (!pw) {
      maai_set_error("conv2d_igemm: the statistics-only and BN-backward epilogues are for pointwise stride-1 layers");
      return MAAI_ERR_UNSUPPORTED;
    }
    return launch_conv_p<T, BM, BN, NSTAGE, EMODE, true>(a, st);
  } else {
    return pw ? launch_conv_p<T, BM, BN, NSTAGE, EMODE, true>(a, st) : launch_conv_p<T, BM, BN, NSTAGE, EMODE, false>(a, st);
  }
}

// the fused-epilogue variants are separate instantiations so that the plain kernel keeps its register budget
// (76 VGPRs -> 3 waves/SIMD); they exist for the 128-row tile only (the fused units are pointwise layers)
template <typename T, int BM, int BN, int NSTAGE>
static int launch_conv_n(const ConvArgs& a, hipStream_t st) {
  if constexpr (BM == 128) {
    switch (a.emode) {
      case 1: return launch_conv_e<T, BM, BN, NSTAGE, 1>(a, st);
      case 2: return launch_conv_e<T, BM, BN, NSTAGE, 2>(a, st);
      case 3: return launch_conv_e<T, BM, BN, NSTAGE, 3>(a, st);
      case 4: return launch_conv_e<T, BM, BN, NSTAGE, 4>(a, st);
      default: break;
    }
  }
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) return launch_conv_e<T, BM, BN, NSTAGE, 6>(a, st);
  if constexpr (BM == 256 && sizeof(T) == 2) {   // the frozen-BatchNorm epilogue also on the channel-reducing pointwise layers' 256-row tile
    if (a.emode == MAAI_EPI_BN_ACT) return launch_conv_e<T, BM, BN, NSTAGE, 2>(a, st);
  }
  if (a.emode != 0) {
    maai_set_error("conv2d_igemm: fused epilogues need the 128-row tile");
    return MAAI_ERR_UNSUPPORTED;
  }
  if (a.accumulate || a.mask) return launch_conv_e<T, BM, BN, NSTAGE, 5>(a, st);
  return launch_conv_e<T, BM, BN, NSTAGE, 0>(a, st);
}

// ring depth by K extent: short K loops (the HBM-bound 1x1 convolutions) never fill a deep ring and are
// better served by more resident workgroups per CU (LDS is what limits them)
template <typename T, int BM, int BN>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  const int kt = a.KH * a.KW * a.Cin * (int)sizeof(T) / 64;
  static const int forced = getenv("MAAI_CONV_NSTAGE") ? atoi(getenv("MAAI_CONV_NSTAGE")) : 0;  // tuning knob
  if (forced == 2) return launch_conv_n<T, BM, BN, 2>(a, st);
  if (forced == 3) return launch_conv_n<T, BM, BN, 3>(a, st);
  if (forced == 4) return launch_conv_n<T, BM, BN, 4>(a, st);
  if (kt <= 2) return launch_conv_n<T, BM, BN, 2>(a, st);
  // measured on MI355X (scripts/conv_micro.py): 3 slots (48 KB -> 3 workgroups per CU) beat 4 on every
  // ResNet-50 layer shape, including the K = 4608 3x3 convolutions (788 vs 770 TFLOP/s)
  return launch_conv_n<T, BM, BN, 3>(a, st);
}

// M-tile rows.  256-row tiles for the long-K spatial layers (bf16): each wave then owns 128x64 (or 64x64 in a 4x1
// wave column for 64-channel outputs), i.e. 6 LDS-DMA + 12 fragment reads per 32 MFMAs instead of 4 + 8 per 16 —
// the K loop is issue-bound on exactly those (profiles/r01_pmc_conv3x3_sq.json).  Measured on MI355X
// (scripts/conv_micro.py, B = 64): 3x3 C128@112 769 -> 803, C256@56 806 -> 877, C512@28 797 -> 912 TFLOP/s.  Not
// for the pointwise layers (HBM-bound; their fused epilogues are 128-row only) nor for launches that would leave
// CUs idle.  MAAI_CONV_BM = 64 | 128 | 256 overrides (read per call so the tests can toggle it); a 64-row variant
// loses on every ResNet-50 shape.
static int choose_bm(const maai_conv_desc* d, int dtype) {
  const char* e = getenv("MAAI_CONV_BM");
  const int forced = e ? atoi(e) : 0;
  if (forced == 64) return 64;
  if (forced == 128 || dtype != MAAI_BF16) return 128;
  if (forced == 256 && (d->KH * d->KW > 1 || d->Cout % 128 == 0)) return 256;
  // pointwise layers: only the channel-reducing ones (Cin >= 2 Cout, long K for few output columns) gain from the
  // taller tile (scripts/conv_pw_ab.py, B = 256: 1024->256 693 -> 768, 2048->512 781 -> 881, 512->128 506 -> 537 TFLOP/s);
  // the expanding, HBM-bound ones lose (64->256: 248 -> 208)
  if (d->KH * d->KW == 1) {
    const long long Mp = (long long)d->N * d->OHg * d->OWg;
    return (d->Cout % 128 == 0 && d->Cin >= 2 * d->Cout && ((Mp + 255) / 256) * (d->Cout / 128) >= 512) ? 256 : 128;
  }
  const long long M = (long long)d->N * d->OHg * d->OWg;
  const long long tiles = ((M + 255) / 256) * (d->Cout / (d->Cout % 128 == 0 ? 128 : 64));
  return (d->KH * d->KW * d->Cin >= 512 && tiles >= 512) ? 256 : 128;
}

// Halo-staged tiles (HALO kernels) for the 3x3 stride-1 same-size layers whose planes the 16x16 patches cover
// without much waste.  MAAI_CONV_HALO = 0 | 1 overrides the shape rule (read per call, for tests and A/B runs).
struct ConvPlan {
  int bm;
  bool halo;
  int tilesX, tilesY;
  long long nMB;
};
static ConvPlan conv_plan(const maai_conv_desc* d, int dtype) {
  ConvPlan p;
  p.bm = choose_bm(d, dtype);
  p.halo = false;
  p.tilesX = p.tilesY = 0;
  const long long M = (long long)d->N * d->OHg * d->OWg;
  const bool shape_ok = dtype == MAAI_BF16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 &&
                        d->OHg == d->IH && d->OWg == d->IW && d->OH == d->OHg && d->OW == d->OWg && d->out_stride == 1 &&
                        d->out_off_h == 0 && d->out_off_w == 0 && d->Cin % 32 == 0;
  if (shape_ok) {
    const char* e = getenv("MAAI_CONV_HALO");
    const int forced = e ? atoi(e) : -1;
    const int tx = (d->OW + 15) / 16, ty = (d->OH + 15) / 16;
    const long long tiles = (long long)d->N * tx * ty * (d->Cout / (d->Cout % 128 == 0 ? 128 : 64));
    const double cover = (double)d->OH * d->OW / ((double)tx * 16 * ty * 16);
    // measured inside bench.py (B = 256, per launch): C64@224 1.72 -> 1.38 ms, C128@112 1.03 -> 0.88 ms; planes the
    // 16x16 patches cover badly lose it again (C256@56, 77 % cover: 0.91 -> 0.96 ms), so those stay row-staged
    const bool pays = cover >= 0.9;
    if (forced == 1 || (forced != 0 && pays && tiles >= 512)) {
      p.halo = true;
      p.bm = 256;
      p.tilesX = tx;
      p.tilesY = ty;
      p.nMB = (long long)d->N * tx * ty;
      return p;
    }
  }
  p.nMB = (M + p.bm - 1) / p.bm;
  return p;
}

template <int BN>
static int launch_halo(const ConvArgs& a, hipStream_t st) {
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) return launch_conv_p<bf16_t, 256, BN, 3, 6, false, true>(a, st);
  if (a.emode == MAAI_EPI_BN_ACT) return launch_conv_p<bf16_t, 256, BN, 3, 2, false, true>(a, st);
  if (a.emode != 0) {
    maai_set_error("conv2d_igemm: BN epilogues are for pointwise layers");
    return MAAI_ERR_UNSUPPORTED;
  }
  if (a.accumulate || a.mask) return launch_conv_p<bf16_t, 256, BN, 3, 5, false, true>(a, st);
  return launch_conv_p<bf16_t, 256, BN, 3, 0, false, true>(a, st);
}

// The streaming kernel (conv_pws.hip) takes the channel-EXPANDING pointwise stride-1 bf16 layers with 64 / 128 / 256
// input channels (conv3 of the bottlenecks of layers 1-3, the stride-1 projection shortcut).  Measured against the ring
// kernel (scripts/pws_ab.py, B = 256, per launch): 64->256@224 1.72 -> 1.63 ms (normalise-on-load 1.99 -> 1.65),
// 128->512@112 1.20 -> 0.96 (1.40 -> 0.97), 256->1024@56 0.79 -> 0.64 (0.94 -> 0.66), 256->512@56 0.39 -> 0.37 (0.48 ->
// 0.39); equal-channel and channel-reducing layers stay on the ring kernel (-2 ... -11 %).  Its statistics slab has one
// row per 128 pixels, so the row count and the launch decide with this same function (a ReLU mask, which the row count
// cannot see, keeps the ring kernel on 128-row tiles).  MAAI_CONV_PWS = 0 (off) | 1 (shape rule, default) | 2 (every
// shape the kernel is built for), read per call.
static bool pws_selected(const maai_conv_desc* d, const maai_conv_epilogue* epi, int dtype) {
  const char* e = getenv("MAAI_CONV_PWS");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  // a forced ring-kernel tile (the parity tests sweep those knobs) means the ring kernel
  if (mode == 1 && (getenv("MAAI_CONV_BM") || getenv("MAAI_CONV_BN") || getenv("MAAI_CONV_NSTAGE"))) return false;
  const bool pw1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
  if (!(dtype == MAAI_BF16 && pw1 && d->Cout % 64 == 0)) return false;
  if (d->out_stride != 1 || d->OH != d->OHg || d->OW != d->OWg) return false;
  if (!(d->Cin == 64 || d->Cin == 128 || d->Cin == 256)) return false;
  if (epi && epi->mode == MAAI_EPI_DGRAD_REDUCE) {
    // the sum-only data-gradient epilogue with a 1-bit mask (the consumer of the gradient is a folded unit), store or
    // accumulate: conv1's data gradient of the identity blocks of stages 1-3 (64 -> 256, 128 -> 512, 256 -> 1024)
    static const bool off = getenv("MAAI_PWS_DGRAD") && atoi(getenv("MAAI_PWS_DGRAD")) == 0;   // A/B knob
    return !off && !epi->t && epi->mask_bits && !epi->xs && !epi->xb && !epi->a2 && !epi->pre_x && !epi->x2 && !epi->bias && !epi->diag &&
           !epi->sum_increment && d->Cout >= 2 * d->Cin;
  }
  if (d->accumulate) return false;
  if (epi && ((epi->mode != MAAI_EPI_STORE && !((epi->mode == MAAI_EPI_STATS_ONLY || epi->mode == MAAI_EPI_BN_ACT) && !epi->xb)) || epi->a2 || epi->pre_x)) return false;
  if (epi && epi->mode == MAAI_EPI_BN_ACT && d->Cout > 1024) return false;   // (its coefficient table lives in the statistics scratch)
  // ... and every forward launch with 256 input channels (conv1 of layer 1's blocks and of layer2.0: 1-7 % slower than the
  // ring kernel there): the chained launch of conv_chain.hip, which replaces the join-on-load form of those launches in
  // forwards without a backward pass, adds the statistics of ITS output in the streaming kernel's order, and plain, joined
  // and chained boundaries must all give the same slab bit for bit
  return mode == 2 || d->Cout >= 2 * d->Cin || d->Cin == 256;
}

// The 8-wave ping-pong kernel (conv_pp.hip) takes the MFMA-bound bf16 layers with >= 256 output channels: the 3x3
// convolutions of stages 3-4 and their data gradients.  Its statistics slab has one row per 256 pixels — the row count
// conv_plan already gives these shapes (256-row tiles), so selecting it changes no slab shape.
// MAAI_CONV_PP = 0 (off) | 1 (shape rule, default) | 2 (every shape the kernel is built for), read per call.
// The persistent resident-weights kernel (conv_c64.hip) takes the 64 -> 64 channel 3x3 stride-1 layers on large planes
// (conv2 of layer 1's bottlenecks): forward with a tensor or a normalise-on-load input, and the data gradient whose mask
// comes from the unit below's y*scale + shift.  One statistics-slab row per WAVE (6 per workgroup, one workgroup per CU).
// MAAI_CONV_C64 = 0 (off) | 1 (shape rule, default) | 2 (every shape it is built for), read per call.
static bool c64_selected(const maai_conv_desc* d, const maai_conv_epilogue* epi, int dtype, const void* relu_mask) {
  const char* e = getenv("MAAI_CONV_C64");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  if (mode == 1 && (getenv("MAAI_CONV_BM") || getenv("MAAI_CONV_BN") || getenv("MAAI_CONV_NSTAGE") || getenv("MAAI_CONV_HALO"))) return false;
  if (dtype != MAAI_BF16 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_h != 1 || d->pad_w != 1 || d->Cin != 64 || d->Cout != 64) return false;
  if (d->OHg != d->IH || d->OWg != d->IW || d->OH != d->OHg || d->OW != d->OWg || d->out_stride != 1 || d->out_off_h || d->out_off_w) return false;
  if (d->accumulate) return false;
  const int emode = epi ? epi->mode : MAAI_EPI_STORE;
  if (epi && (epi->xb || epi->a2 || epi->pre_x)) return false;
  if (emode == MAAI_EPI_DGRAD_REDUCE) {
    // (any mask form — none, tensor, 1-bit, or the unit below's y*scale + shift: the slab's row count is asked for without
    //  knowing the mask, so the choice must not depend on it)
    if (!epi->t || epi->xs) return false;
  } else if (emode == MAAI_EPI_STORE) {
    if (relu_mask) return false;   // (a masked store writes no slab)
  } else {
    return false;
  }
  if (mode == 2) return true;
  const long long patches = (long long)d->N * ((d->IW + 15) / 16) * ((d->IH + 6) / 7);
  return patches >= 8192 && d->IW % 16 == 0 && d->IH >= 96;   // (16 x 6 patches: a ragged last row of patches costs < 6 % from 96 rows up)
}
static long long c64_rows(const maai_conv_desc* d) {
  ConvArgs a;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW;
  return maai_conv_c64_rows(a);
}

static bool pp_forced() {
  const char* e = getenv("MAAI_CONV_PP");
  return e && atoi(e) == 2;
}
static bool pp_selected(const maai_conv_desc* d, const maai_conv_epilogue* epi, int dtype, const void* relu_mask) {
  const char* e = getenv("MAAI_CONV_PP");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  if (mode == 1 && (getenv("MAAI_CONV_BM") || getenv("MAAI_CONV_BN") || getenv("MAAI_CONV_NSTAGE") || getenv("MAAI_CONV_HALO"))) return false;
  const int emode = epi ? epi->mode : MAAI_EPI_STORE;
  if (dtype != MAAI_BF16 || d->Cin % 64 || d->Cout % 128 || d->KH * d->KW > 32 || d->KH * d->KW * d->Cin < 128) return false;
  if (d->out_stride != 1 || d->out_off_h || d->out_off_w || d->OH != d->OHg || d->OW != d->OWg) return false;
  if ((long long)d->N * d->IH * d->IW * d->Cin >= (1ll << 31)) return false;
  if (epi && (epi->xs || epi->xb || epi->a2 || epi->pre_x)) return false;
  if (emode == MAAI_EPI_STORE) {
    if (d->accumulate || relu_mask) return false;
  } else if (emode == MAAI_EPI_BN_ACT) {
    if (d->accumulate || relu_mask || d->Cout % 256) return false;
  } else if (emode != MAAI_EPI_DGRAD_REDUCE) {
    return false;
  }
  if (mode == 2) return true;
  // measured against the ring / halo / streaming kernels (scripts/pp_ab.py, 256 images, interleaved, bit-identical outputs):
  //   3x3: 256->256@56 1.13 -> 0.94 ms, 512->512@28 0.98 -> 0.80, stride 2 256@112 1.12 -> 0.89, 512@56 1.02 -> 0.81;
  //   data gradients (mask + sums epilogue) 256@56 1.36 -> 1.17, 512@28 1.12 -> 0.96;
  //   1x1: 1024->256@56 0.57 -> 0.48, 2048->512@28 0.50 -> 0.41, 1024->512@56 1.02 -> 0.88, 1024->2048@28 0.95 -> 0.84,
  //        512->256@112 1.30 -> 1.16; the channel-expanding ones lose (256->1024@56: 0.64 -> 0.81, streaming kernel) or tie
  //        (512->2048@28 0.57 -> 0.56).
  const char* r = getenv("MAAI_CONV_PP_RULE");   // experiment knob: bit 0 = 3x3 >= 256 channels, 1 = long-K 1x1, 2 = 128-channel 3x3, 3 = 1:4 expanding 1x1 with the BN_ACT epilogue (eval forward 55.0 -> 54.45 ms)
  const int rule = r ? atoi(r) : 11;
  if (d->KH * d->KW >= 9) return d->Cout % 256 == 0 ? (rule & 1) && d->Cin >= 256 : (rule & 4) && d->Cin >= 128;
  // (rule bit 3: with the frozen-BatchNorm epilogue also the 1:4 expanding layers — 512->2048@28 — whose plain launch ties
  //  with the ring kernel's 128 x 256 tile, but whose fused epilogue there is slow)
  if (d->KH * d->KW == 1)
    return (rule & 2) && d->Cout % 256 == 0 && d->Cin >= 512 &&
           (d->Cout <= 2 * d->Cin || ((rule & 8) && emode == MAAI_EPI_BN_ACT && d->Cout <= 4 * d->Cin));
  return false;
}

extern "C" int maai_conv2d_igemm(const maai_conv_desc* d, const void* x, const void* w, void* y, float* stats_partial,
                                 const void* relu_mask, int dtype, void* stream) {
  return maai_conv2d_igemm_fused(d, x, w, y, stats_partial, relu_mask, nullptr, dtype, stream);
}

extern "C" int maai_conv2d_igemm_fused(const maai_conv_desc* d, const void* x, const void* w, void* y, float* stats_partial,
                                       const void* relu_mask, const maai_conv_epilogue* epi, int dtype, void* stream) {
  const int emode = epi ? epi->mode : MAAI_EPI_STORE;
  MAAI_CHECK_ARG(d && w && (x || (epi && epi->pre_x)), "conv2d_igemm: null pointer");
  MAAI_CHECK_ARG(emode >= 0 && emode <= MAAI_EPI_DGRAD_REDUCE, "conv2d_igemm: bad epilogue mode");
  MAAI_CHECK_ARG(y || emode == MAAI_EPI_STATS_ONLY || emode == MAAI_EPI_BWD_REDUCE, "conv2d_igemm: null output");
  MAAI_CHECK_ARG((emode != MAAI_EPI_STATS_ONLY && emode != MAAI_EPI_BWD_REDUCE && emode != MAAI_EPI_DGRAD_REDUCE) || stats_partial, "conv2d_igemm: this epilogue needs the partial-sum slab");
  MAAI_CHECK_ARG(emode < MAAI_EPI_BWD_REDUCE || epi->t || (emode == MAAI_EPI_DGRAD_REDUCE && relu_mask),
                 "conv2d_igemm: BN-backward epilogues need dz (DGRAD_REDUCE: the raw conv output, or — first sum only — a mask)");
  MAAI_CHECK_ARG(emode != MAAI_EPI_BWD_APPLY || (epi->p0 && epi->p1 && epi->p2), "conv2d_igemm: BN-backward apply needs k1, k2, k3");
  MAAI_CHECK_ARG(emode < 2 || emode == MAAI_EPI_DGRAD_REDUCE || (!d->accumulate && !relu_mask && d->out_stride == 1), "conv2d_igemm: fused BN epilogues are dense, non-accumulating");
  MAAI_CHECK_ARG(dtype == MAAI_BF16 || dtype == MAAI_F32, "conv2d_igemm: dtype must be MAAI_BF16 or MAAI_F32");
  const int bk = dtype == MAAI_BF16 ? 32 : 16;
  MAAI_CHECK_ARG(d->N > 0 && d->IH > 0 && d->IW > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0, "conv2d_igemm: bad dims");
  MAAI_CHECK_ARG(d->Cin % bk == 0, "conv2d_igemm: Cin must be a multiple of 32 (bf16) / 16 (f32)");
  MAAI_CHECK_ARG(d->Cout % 64 == 0, "conv2d_igemm: Cout must be a multiple of 64");
  MAAI_CHECK_ARG(d->OHg > 0 && d->OWg > 0 && d->out_stride >= 1, "conv2d_igemm: bad output grid");
  MAAI_CHECK_ARG((d->OHg - 1) * d->out_stride + d->out_off_h < d->OH && (d->OWg - 1) * d->out_stride + d->out_off_w < d->OW,
                 "conv2d_igemm: output scatter exceeds the output tensor");
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.stats = stats_partial; a.mask = relu_mask;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
  a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.OHg = d->OHg; a.OWg = d->OWg;
  a.OH = d->OH; a.OW = d->OW; a.ostr = d->out_stride; a.ooh = d->out_off_h; a.oow = d->out_off_w;
  a.accumulate = d->accumulate;
  a.emode = emode;
  a.erelu = epi ? epi->relu : 0;
  a.ep0 = epi ? epi->p0 : nullptr;
  a.ep1 = epi ? epi->p1 : nullptr;
  a.ep2 = epi ? epi->p2 : nullptr;
  a.et = epi ? epi->t : nullptr;
  a.mask_bits = epi ? epi->mask_bits : 0;
  a.sum_incr = epi ? epi->sum_increment : 0;
  a.a2 = epi ? epi->a2 : nullptr;
  a.ak1 = epi ? epi->ak1 : nullptr;
  a.ak2 = epi ? epi->ak2 : nullptr;
  a.ak3 = epi ? epi->ak3 : nullptr;
  a.a_out = epi ? epi->a_out : nullptr;
  a.xs = epi ? epi->xs : nullptr;
  a.xt = epi ? epi->xt : nullptr;
  a.x_relu = epi ? epi->x_relu : 0;
  a.xb = epi ? epi->xb : nullptr;
  a.xs2 = epi ? epi->xs2 : nullptr;
  a.xt2 = epi ? epi->xt2 : nullptr;
  a.x_out = epi ? epi->x_out : nullptr;
  a.x_bits = epi ? epi->x_bits : nullptr;
  a.pre_x = epi ? epi->pre_x : nullptr;
  a.pre_w = epi ? epi->pre_w : nullptr;
  a.pre_xs = epi ? epi->pre_xs : nullptr;
  a.pre_xt = epi ? epi->pre_xt : nullptr;
  a.pre_relu = epi ? epi->pre_relu : 0;
  a.pre_cin = epi ? epi->pre_cin : 0;
  a.pre_y_out = epi ? epi->pre_y_out : nullptr;
  a.x2 = epi ? epi->x2 : nullptr;
  a.cin1 = epi ? epi->cin1 : 0;
  a.ebias = epi ? epi->bias : nullptr;
  a.ediag = epi ? epi->diag : nullptr;
  MAAI_CHECK_ARG(!a.ediag || (emode == MAAI_EPI_DGRAD_REDUCE && epi->t && epi->p1 && epi->p2),
                 "conv2d_igemm: diag needs the DGRAD_REDUCE epilogue with the lower unit's raw output, scale and shift");
  if (a.x2 || a.ebias) {
    const bool pw1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
    MAAI_CHECK_ARG(emode == MAAI_EPI_DGRAD_REDUCE && dtype == MAAI_BF16 && pw1 && !a.xs && !a.a2 && !a.pre_x,
                   "conv2d_igemm: the two-source input and the bias belong to bf16 pointwise DGRAD_REDUCE launches");
    MAAI_CHECK_ARG(!a.x2 || (a.cin1 > 0 && a.cin1 < d->Cin && a.cin1 % 64 == 0 && (d->Cin - a.cin1) % 64 == 0),
                   "conv2d_igemm: the two-source input splits Cin into two multiples of 64");
  }
  const bool pws = pws_selected(d, epi, dtype);
  if (a.x_bits && !a.xb) {
    MAAI_CHECK_ARG(emode == MAAI_EPI_BN_ACT && pws && !a.xs && a.erelu && dtype == MAAI_BF16,
                   "conv2d_igemm: x_bits without a two-tensor join is the mask output of the streaming kernel's BatchNorm + ReLU epilogue");
  }
  if (a.xs || a.xb) {
    MAAI_CHECK_ARG(a.xs && a.xt && (emode == MAAI_EPI_STORE || ((emode == MAAI_EPI_STATS_ONLY || (emode == MAAI_EPI_BN_ACT && !a.xb)) && pws)) && !d->accumulate && !relu_mask &&
                       d->out_stride == 1 && !a.a2,
                   "conv2d_igemm: the normalised-on-load operand needs xs and xt and a plain dense forward launch");
    const bool pw1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
    MAAI_CHECK_ARG(!a.xb || pw1, "conv2d_igemm: the two-tensor join on load is for pointwise stride-1 layers");
    MAAI_CHECK_ARG((a.xs2 == nullptr) == (a.xt2 == nullptr) && (a.xb || (!a.xs2 && !a.x_out && !a.x_bits)),
                   "conv2d_igemm: xs2/xt2 come in pairs and, like x_out/x_bits, belong to the two-tensor join");
    MAAI_CHECK_ARG(!a.x_bits || (a.x_out && dtype == MAAI_BF16), "conv2d_igemm: the 1-bit mask of the joined activation is for bf16 x_out");
  }
  if (a.pre_x) {
    const bool pw1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
    MAAI_CHECK_ARG(a.pre_w && a.xb && a.xs && a.xt && a.x_out && y && dtype == MAAI_BF16 && pw1 && emode == MAAI_EPI_STORE &&
                       (a.pre_xs == nullptr) == (a.pre_xt == nullptr) && d->OH == d->OHg && d->OW == d->OWg,
                   "conv2d_igemm: a chained launch is a bf16 pointwise two-tensor join with pre_w and x_out");
    MAAI_CHECK_ARG(a.pre_cin == 64 && d->Cin == 256, "conv2d_igemm: the chained launch is built for pre_cin 64, Cin 256");
  }
  const bool axf = a.a2 != nullptr;
  if (axf) {
    const bool pw1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
    MAAI_CHECK_ARG(emode == MAAI_EPI_DGRAD_REDUCE && dtype == MAAI_BF16 && pw1 && a.ak1 && a.ak2 && a.ak3 && d->Cin <= 4096,
                   "conv2d_igemm: the transformed A operand is for bf16 pointwise DGRAD_REDUCE launches with k1, k2, k3");
  }
  MAAI_CHECK_ARG(!a.mask_bits || (emode == MAAI_EPI_DGRAD_REDUCE && dtype == MAAI_BF16 && relu_mask),
                 "conv2d_igemm: the 1-bit mask is for the bf16 DGRAD_REDUCE epilogue");
  a.M = (long long)d->N * d->OHg * d->OWg;
  MAAI_CHECK_ARG(a.M < (1ll << 31), "conv2d_igemm: pixel count must fit 31 bits");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ConvPlan plan = conv_plan(d, dtype);
  const bool pw1x1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
  if (axf || (emode >= MAAI_EPI_STATS_ONLY && emode <= MAAI_EPI_BWD_APPLY &&
              !(emode == MAAI_EPI_BN_ACT && (plan.halo || (plan.bm == 256 && pw1x1 && dtype == MAAI_BF16))))) {
    // 128-row, row-staged tiles only (the frozen-BatchNorm epilogue also exists on the halo, streaming and ping-pong kernels)
    plan.bm = 128;
    plan.halo = false;
    plan.nMB = (a.M + 127) / 128;
  }
  const bool xf = a.xs != nullptr;
  if (xf && plan.bm == 64) {  // (a tuning knob's tile the transformed-operand kernels are not built for)
    plan.bm = 128;
    plan.nMB = (a.M + 127) / 128;
  }
  a.nMB = (int)plan.nMB;
  a.tilesX = plan.tilesX;
  a.tilesY = plan.tilesY;
  ConvSel sel;
  sel.dtype = dtype; sel.bm = plan.bm; sel.halo = plan.halo; sel.nstage = 3;
  sel.pw = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
  if (a.pre_x) return maai_conv_chain_launch(a, st);
  if (c64_selected(d, epi, dtype, relu_mask)) {
    if (!maai_conv_c64_supported(a, dtype)) {
      maai_set_error("conv2d_igemm: internal: the 64-channel 3x3 kernel's shape rule and its support test disagree");
      return MAAI_ERR_UNSUPPORTED;
    }
    return maai_conv_c64_launch(a, st);
  }
  // (the default rule's shapes are never the streaming kernel's; MAAI_CONV_PP=2 — tests, A/B runs — takes every shape it can)
  if (pp_selected(d, epi, dtype, relu_mask) && (!pws || pp_forced())) {
    if (!maai_conv_pp_supported(a, dtype)) {
      maai_set_error("conv2d_igemm: internal: the ping-pong kernel's shape rule and its support test disagree");
      return MAAI_ERR_UNSUPPORTED;
    }
    return maai_conv_pp_launch(a, st);
  }
  if (pws) {
    if (!relu_mask || emode == MAAI_EPI_DGRAD_REDUCE) return maai_conv_pws_launch(a, st);
    plan.bm = 128;   // the slab rows promised for this shape (a masked launch never carries a join: 128-row tiles)
    sel.bm = 128;
    plan.nMB = (a.M + 127) / 128;
    a.nMB = (int)plan.nMB;
  }
  const int bm = plan.bm;
  if (plan.halo) {
    const bool h128 = d->Cout % 128 == 0;
    a.nNB = d->Cout / (h128 ? 128 : 64);
    sel.bn = h128 ? 128 : 64;
    if (xf) return maai_conv_xf_launch(a, sel, st);
    return h128 ? launch_halo<128>(a, st) : launch_halo<64>(a, st);
  }
  static const int force_bn = getenv("MAAI_CONV_BN") ? atoi(getenv("MAAI_CONV_BN")) : 0;  // experiment knob
  const bool n128 = d->Cout % 128 == 0 && !(force_bn == 64 && d->KH * d->KW * d->Cin <= 128);
  a.nNB = d->Cout / (n128 ? 128 : 64);
  // 128x256 tiles for the channel-EXPANDING 1x1 layers: their A operand is re-read from L2 once per column tile
  // and L2 -> LDS staging tops out near 8 TB/s (scripts/probes/dma_probe2.hip), so halving the column tiles halves
  // the staged bytes; the channel-reducing ones get the taller 256x128 tile for the same reason (choose_bm).
  {
    const char* e = getenv("MAAI_CONV_BN");
    const int fbn = e ? atoi(e) : 0;
    if (!axf && dtype == MAAI_BF16 && bm == 128 && d->KH * d->KW == 1 && d->Cout % 256 == 0 && fbn != 128 && fbn != 64 &&
        (emode == MAAI_EPI_STORE || emode == MAAI_EPI_DGRAD_REDUCE || emode == MAAI_EPI_BN_ACT) &&
        (fbn == 256 || ((emode == MAAI_EPI_STORE || emode == MAAI_EPI_BN_ACT) && !d->accumulate && !relu_mask && d->Cout >= 2 * d->Cin && d->Cin >= 256 &&
                        a.nMB * (long long)(d->Cout / 256) >= 512))) {
      // measured in bench.py (B = 256, per launch): 256->1024 0.78 -> 0.74 ms, 512->2048 0.59 -> 0.56, 1024->2048/s2
      // 1.09 -> 0.96; the read-modify-write epilogues (3.3 -> 4.1 ms on 128->512 with accumulate + sums) and K <= 128
      // lose with the wider tile, so the shape rule covers the plain forward layers only
      a.nNB = d->Cout / 256;
      sel.bn = 256;
      if (xf) return maai_conv_xf_launch(a, sel, st);
      return launch_conv_n<bf16_t, 128, 256, 3>(a, st);
    }
  }
  if (xf) {
    sel.bn = n128 ? 128 : 64;
    sel.nstage = (bm == 128 && d->KH * d->KW * d->Cin * (dtype == MAAI_BF16 ? 2 : 4) / 64 <= 2) ? 2 : 3;
    return maai_conv_xf_launch(a, sel, st);
  }
  if (axf) {
    return n128 ? launch_conv_p<bf16_t, 128, 128, 2, 6, true, false, true>(a, st) : launch_conv_p<bf16_t, 128, 64, 2, 6, true, false, true>(a, st);
  }
  if (dtype == MAAI_BF16) {
    if (bm == 64) return n128 ? launch_conv<bf16_t, 64, 128>(a, st) : launch_conv<bf16_t, 64, 64>(a, st);
    if (bm == 256) return n128 ? launch_conv_n<bf16_t, 256, 128, 3>(a, st) : launch_conv_n<bf16_t, 256, 64, 3>(a, st);
    return n128 ? launch_conv<bf16_t, 128, 128>(a, st) : launch_conv<bf16_t, 128, 64>(a, st);
  }
  if (bm == 64) return n128 ? launch_conv<float, 64, 128>(a, st) : launch_conv<float, 64, 64>(a, st);
  return n128 ? launch_conv<float, 128, 128>(a, st) : launch_conv<float, 128, 64>(a, st);
}

/* rows of the statistics slab: one per M-tile of the kernel maai_conv2d_igemm picks for (d, dtype) */
/* the same for a launch with an epilogue descriptor (the transformed-operand launches use 128-row tiles) */
extern "C" long long maai_conv2d_stats_rows_fused(const maai_conv_desc* d, const maai_conv_epilogue* epi, int dtype) {
  if (!d) return 0;
  if (epi && (epi->a2 || epi->pre_x || (epi->mode >= MAAI_EPI_STATS_ONLY && epi->mode <= MAAI_EPI_BWD_APPLY)))
    return ((long long)d->N * d->OHg * d->OWg + 127) / 128;
  const ConvPlan p = conv_plan(d, dtype);
  if (epi && epi->xs && p.bm == 64) return ((long long)d->N * d->OHg * d->OWg + 127) / 128;
  if (c64_selected(d, epi, dtype, nullptr)) return c64_rows(d);
  // (the ping-pong kernel: one row per 256 pixels; a relu mask is unknown here — masked STORE launches write no slab)
  const bool pp = pp_selected(d, epi, dtype, nullptr);
  const long long pprows = maai_conv_pp_rows(d->Cout);
  if (pp && pp_forced()) return ((long long)d->N * d->OHg * d->OWg + pprows - 1) / pprows;
  if (pws_selected(d, epi, dtype)) return ((long long)d->N * d->OHg * d->OWg + 127) / 128;
  if (pp) return ((long long)d->N * d->OHg * d->OWg + pprows - 1) / pprows;
  return p.nMB;
}

/* which kernel family a PLAIN forward launch (no epilogue descriptor, tensor input) of this geometry gets:
 * 0 = ring / halo (conv_igemm.h), 1 = streaming (conv_pws.hip), 2 = ping-pong (conv_pp.hip), 3 = persistent 64-channel 3x3
 * (conv_c64.hip, which also takes normalise-on-load inputs).  Each family sums the BatchNorm
 * partial statistics in its own order; the engine keeps a layer on ONE family whatever form its input has. */
extern "C" int maai_conv2d_kernel_family(const maai_conv_desc* d, int dtype) {
  if (!d) return 0;
  if (c64_selected(d, nullptr, dtype, nullptr)) return 3;
  const bool pp = pp_selected(d, nullptr, dtype, nullptr);
  if (pp && pp_forced()) return 2;
  if (pws_selected(d, nullptr, dtype)) return 1;
  return pp ? 2 : 0;
}

/* Does a MAAI_EPI_BN_ACT launch (frozen statistics) of this geometry run on the kernel its plain launch would use — the
 * streaming, ping-pong or halo kernel, or the ring kernel's natural 128-row tile?  ``lazy``: with a normalise-on-load input
 * (the streaming kernel only).  Otherwise it falls back to 128-row row-staged tiles, slower than launch + pass on large tensors. */
extern "C" int maai_conv2d_bn_act_fast(const maai_conv_desc* d, int dtype, int lazy) {
  if (!d) return 0;
  maai_conv_epilogue e;
  memset(&e, 0, sizeof(e));
  e.mode = MAAI_EPI_BN_ACT;
  if (pws_selected(d, &e, dtype)) return 1;
  if (lazy) return 0;
  if (pp_selected(d, &e, dtype, nullptr)) return 1;
  const ConvPlan p = conv_plan(d, dtype);
  const bool pw1x1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OHg == d->IH && d->OWg == d->IW;
  return (p.halo || p.bm == 128 || (p.bm == 256 && pw1x1 && dtype == MAAI_BF16)) ? 1 : 0;
}

extern "C" long long maai_conv2d_stats_rows(const maai_conv_desc* d, int dtype) {
  return maai_conv2d_stats_rows_fused(d, nullptr, dtype);
}
