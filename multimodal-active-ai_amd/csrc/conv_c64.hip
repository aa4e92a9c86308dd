// 3x3 stride-1 convolution with 64 input and 64 output channels at large planes (bf16): conv2 of layer 1's bottlenecks at
// 224^2 (resnet.py:107, conv3x3(width, width)), forward (plain or normalise-on-load input) and data gradient (mask of the
// unit below + BatchNorm-backward sums).
//
// These launches are HBM-bound on paper (3.3 GB forward, 4.9 GB data gradient at 224^2 x 256 images: ~0.6 / 0.9 ms) with
// 0.95 TFLOP of MFMA work (~0.4 ms at peak): loads, MFMAs and epilogues have to OVERLAP, and the halo kernel of
// conv_igemm.h — 18 barrier-separated K-steps per 256-row tile, two workgroups per CU, prologue and epilogue exposed —
// runs them at 1.36 / 1.79 / 1.99 ms inside the step.  Here:
//   * the weights (64 x 9 x 64 bf16 = 72 KB) stay in LDS for the whole launch;
//   * one persistent workgroup of FOUR waves per CU — one per SIMD, up to 512 registers each; every WAVE owns its own stream
//     of 16 x 7-pixel output patches (16 x 6 in the data gradient) and its own 18 x 9-pixel halo image in LDS (20 KB): no
//     workgroup barrier after the weights are in, no SIMD carries more waves than another (a first version had six waves of
//     16 x 4 patches, all that LDS admits at 256 registers: two SIMDs carried two waves, two carried one — 0.99 / 1.20 / 1.85 ms);
//   * the next patch's halo is loaded into REGISTERS (21 x 16 bytes per lane: row segments of 1 KB, two edge columns)
//     before the current patch's 504 MFMAs and written to LDS at the top of the next iteration;
//   * the weights are the MFMA's A operand, permuted so that one lane ends up with EIGHT CONSECUTIVE output channels of a
//     pixel: the epilogue is 16-byte row stores straight from the accumulators — no LDS transpose (the very first version
//     had one: 256 ds_write_b16 per patch, 1.22 ms);
//   * the packed output rows are stored one iteration LATE, in front of the request for the halo after next: issued right
//     after the MFMAs they are younger than the halo loads in flight, and the waits hipcc derives for the LDS refill (it
//     counts the loads only) then run down to vmcnt(0) — every refill waited for the previous patch's stores to be
//     acknowledged (data gradient 2.16 -> 2.00 ms);
//   * fragments are double-buffered by hand (the reads of K-step k + 1 are issued before the MFMAs of step k);
//   * normalise-on-load (XF): applied once per halo element on the way from the prefetch registers to LDS, i.e. a whole
//     patch after the loads were issued (applied at load time it waits for them: 1.94 instead of 1.20 ms); padding stays zero;
//   * BatchNorm statistics (forward) / BatchNorm-backward sums (data gradient) accumulate in registers over ALL patches
//     of a wave: one slab row per wave (deterministic: patches are dealt round-robin).  A wave's fp32 sums therefore run
//     over ~110 patches where a tile kernel's run over 16 rows: totals of two launches that split the batch differently
//     agree to fp32 summation order (3e-5 relative), not bit for bit.
// K order = the halo kernel's (32-channel chunk major, tap minor, one MFMA per output tile and step): outputs are
// bit-identical to it (tests/test_gpu_c64.py).  Halo image: [9 rows][18 pixels][128 B], chunk ^= c64_key(hx) (below: conflict-
// free for the lane groups ds_read_b128 is really served in), and a halo row is a compile-time offset from three per-lane base addresses (one per kw).
// Measured (B = 256, 224^2, scripts/pp_ab.py c64, medians, halo kernel -> this): forward 1.57 -> 1.0 ms, normalise-on-load
// 1.8 -> 1.1, data gradient (with the 0.6 ms clone the script adds to both arms) 2.63 -> 2.00; the step 808 -> 821 images/s.
// What bounds it as built: each wave moves 35 KB per patch in 8.9 us — 4.0 TB/s over the chip — with the MFMA pipe 43 % busy:
// neither roof.  Ablation builds (six-wave version, one box): compute side alone 0.79 ms, memory side alone 0.77-0.88 ms,
// together 1.1: one wave per SIMD overlaps its own loads with its MFMAs, but its LDS refill, epilogue and scalar patch
// arithmetic run with the matrix pipe idle.  Tried and dropped: requesting the data-gradient epilogue's operands before the K
// loop (spills at 256 registers: 1.85 -> 2.15 ms), halo loads from inline assembly with a hand-counted s_waitcnt (hipcc copies
// the destination registers before the wait: wrong results — the deferred stores above are the compiler-visible way to the
// same end), plain instead of non-temporal halo loads and non-temporal output stores (MAAI_EXP 16 / 32 builds, interleaved
// A/B: all within +-2 %).  The halo swizzle below removed every LDS bank conflict (SQ_LDS_BANK_CONFLICT 25.8 % -> 0.0 % of the
// LDS cycles) and 0.3 % of the time: the kernel runs at ~4 TB/s of HBM traffic (PMC: 3.88 GB per forward launch against 3.29
// algorithmic — part of the halo overlap is fetched twice), 70 % of what the streaming BatchNorm passes reach.
#include "conv_igemm.h"
#include <type_traits>

namespace {
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<N, I + 1>(f);
  }
}
// Halo swizzle: the 16-byte chunk c of halo pixel hx lives at chunk slot c ^ c64_key(hx) of the pixel's 128 bytes.
// ds_read_b128 is served in four groups of sixteen lanes that are NOT lane-contiguous — {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ... (MI355X_MICROARCH.md, LDS): a group reads pixels frow in {0-3, 12-15} at one K-chunk and frow in {4-11} at the
// NEXT one.  The textbook key (hx >> 1) & 7 is conflict-free for sixteen consecutive pixels at ONE chunk; with the real
// groups it collides for the taps kw = 1, 2 (two of sixteen slots double: 25.8 % of the kernel's LDS cycles were bank
// conflicts, SQ_LDS_BANK_CONFLICT).  This key — bit 1 = hx bit 2, bit 2 = hx bit 1 — is conflict-free for kw = 0, 1, 2
// and both group shapes (exhaustive search over the GF(2)-linear keys, scripts in DESIGN.md section 4).
#if MAAI_EXP & 16   // A/B builds (scripts/build_variant.sh): plain instead of non-temporal halo loads / non-temporal output stores
#define C64_LD(p) (*reinterpret_cast<const uint4*>(p))
#else
#define C64_LD(p) ld16_nt(p)
#endif
#if MAAI_EXP & 32
#define C64_ST(p, v) st16_nt(p, v)
#else
#define C64_ST(p, v) (*reinterpret_cast<uint4*>(p) = (v))
#endif
__device__ __forceinline__ constexpr int c64_key(int hx) { return (((hx >> 2) & 1) << 1) | (((hx >> 1) & 1) << 2); }
constexpr int C64_NW = 4;                 // waves per workgroup: ONE per SIMD, up to 512 registers each
// output rows per patch: 16 x 7 pixels (seven MFMA pixel tiles; 224 = 32 x 7) in the forward kernels, 16 x 6 in the data
// gradient, whose epilogue needs the registers (16 x 7 there: 55 spilled)
constexpr int c64_pr(int emode) { return emode == 6 ? 6 : 7; }
constexpr int C64_HP = 18;                // halo pitch (pixels)
constexpr int C64_WB = 9 * 64 * 128;      // 73728 bytes of weights
constexpr int c64_hb(int pr) { return C64_HP * (pr + 2) * 128; }   // bytes of one halo image: (pr + 2) rows x 18 pixels x 128 B
}  // namespace

template <int XF, int EMODE>
__global__ __launch_bounds__(64 * C64_NW, 1) void conv_c64_kernel(ConvArgs a) {
  constexpr int C64_PR = c64_pr(EMODE), C64_HR = C64_PR + 2, C64_HB = c64_hb(C64_PR);
  constexpr int C64_NE = (C64_HR + 3) / 4;           // edge-column loads (four halo rows x two pixels each)
  constexpr int C64_NCH = 2 * C64_HR + C64_NE;       // 16-byte chunks per lane per halo
  typedef bf16_t T;
  typedef Mma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wsm = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* hb = smem + C64_WB + wid * C64_HB;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  const int H = a.IH, W = a.IW;
  const int tilesX = (W + 15) >> 4, tilesY = (H + C64_PR - 1) / C64_PR, tpi = tilesX * tilesY;
  const int P = a.N * tpi;
  // wave index over the launch.  Workgroups go round-robin over the 8 XCDs (each with its own L2): the waves of one XCD
  // take CONTIGUOUS patches in every round, so that the halo rows and columns neighbouring patches share meet in one L2.
  int gw = blockIdx.x * C64_NW + wid;
  const int gstride = gridDim.x * C64_NW;
  if ((gridDim.x & 7) == 0) gw = ((int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)) * C64_NW + wid;

  // ---- weights -> LDS: [tap][row][64 channels], chunk ^= (row >> 1) & 7.  The weights are the MFMA's A operand (rows =
  // output channels): row t*16 + m of the image holds channel 32 (t >> 1) + 8 (m >> 2) + 4 (t & 1) + (m & 3), so that the
  // accumulators of tiles 2q, 2q+1 in one lane are EIGHT CONSECUTIVE channels of one pixel: the epilogue is 16-byte
  // row stores straight from the registers ----
  for (int c = tid; c < 9 * 64 * 8; c += 64 * C64_NW) {
    const int ch = c & 7, row = c >> 3;            // row = tap*64 + t*16 + m
    const int tap = row >> 6, t = (row >> 4) & 3, m = row & 15;
    const int co = 32 * (t >> 1) + 8 * (m >> 2) + 4 * (t & 1) + (m & 3);
    const uint4 v = *reinterpret_cast<const uint4*>(w + ((long long)co * 9 + tap) * 64 + ch * 8);
    *reinterpret_cast<uint4*>(wsm + row * 128 + ((ch ^ ((m >> 1) & 7)) << 4)) = v;
  }
  // per-channel coefficient tables (XF: xs | xt; data gradient: mean | scale | shift) live in LDS behind the halo images
  float* ctab = reinterpret_cast<float*>(smem + C64_WB + C64_NW * C64_HB);
  if constexpr (XF != 0) {
    if (tid < 128) ctab[tid] = tid < 64 ? a.xs[tid] : a.xt[tid - 64];
  }
  if constexpr (EMODE == 6) {
    if (tid < 192)
      ctab[128 + tid] = tid < 64 ? (a.ep0 ? a.ep0[tid] : 0.f) : (tid < 128 ? (a.ep1 ? a.ep1[tid - 64] : 0.f) : (a.ep2 ? a.ep2[tid - 128] : 0.f));
  }
  __syncthreads();

  // ---- halo staging: 12 "body" loads (halo row hy, pixels 8g .. 8g+7: lane = pixel*8 + chunk, 1 KB contiguous) and 2
  // "edge" loads (pixels 16, 17 of four rows at a time) per patch ----
  const int lx = lane >> 3, lch = lane & 7;
  const int sb0 = lx * 128 + ((lch ^ c64_key(lx)) << 4);                       // body (c64_key(8 g + lx) == c64_key(lx)); g = 1: + 1024
  const int er = lane >> 4, ej = (lane >> 3) & 1;
  const int sbe = er * (C64_HP * 128) + (16 + ej) * 128 + (lch << 4);         // edge ((16 + j) >> 1 & 7 == 0: no swizzle)
  const int geoff = (er * W + 16 + ej) * 64 + lch * 8;                         // edge: element offset from the halo origin
  float xs8[XF != 0 ? 8 : 1], xt8[XF != 0 ? 8 : 1];
  if constexpr (XF != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xs8[e] = ctab[lch * 8 + e];
      xt8[e] = ctab[64 + lch * 8 + e];
    }
  }

  // ---- fragment addresses: pixels (B operand) at halo row r + kh, pixel kw + (lane & 15), chunk s*4 + (lane >> 4) ----
  const int frow = lane & 15, fg = lane >> 4;
  int aoff[3];           // [kw]: byte offset in halo row 0, K-chunk 0 (row hy: + hy*2304; chunk 1: ^ 64)
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int hx = kw + frow;
    aoff[kw] = hx * 128 + ((fg ^ c64_key(hx)) << 4);
  }
  const char* wbs[2];    // weights (A operand): image row t*16 + frow, chunk s*4 + fg
#pragma unroll
  for (int s = 0; s < 2; ++s) wbs[s] = wsm + frow * 128 + (((s * 4 + fg) ^ ((frow >> 1) & 7)) << 4);

  // statistics / sums over all patches of this wave: entry q*8 + e <-> channel 32 q + 8 fg + e, pixels (lane & 15) + 16 r
  float s1[16], s2[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;

  uint4 pre[C64_NCH];
  unsigned prem = 0;     // XF: bit i = chunk i of `pre` is inside the image (padding must stay zero through the transform)
  auto patch_origin = [&](int p, int& n, int& oy0, int& ox0) {
    n = p / tpi;
    const int rem = p - n * tpi;
    const int ty = rem / tilesX;
    oy0 = ty * C64_PR;
    ox0 = (rem - ty * tilesX) * 16;
  };
  auto xform = [&](uint4 v, bool ok) -> uint4 {
    if constexpr (XF != 0) {   // act(x*s + t), maai_bn_act_fwd arithmetic; padding stays zero
      Vec16<T> t, u;
      t.raw = v;
      XfMath<T>::template run<false>(t, u, xs8, xt8, nullptr, nullptr, a.x_relu, false);
      return ok ? t.raw : make_uint4(0, 0, 0, 0);
    } else {
      return v;
    }
  };
  auto load_halo = [&](int p) {
    int n, oy0, ox0;
    patch_origin(p, n, oy0, ox0);
    // halo origin = input pixel (oy0 - 1, ox0 - 1); rows are uniform, columns per lane
    const T* org = x + (((long long)n * H + (oy0 - 1)) * W + (ox0 - 1)) * 64;
    if (oy0 >= 1 && oy0 + C64_PR + 1 <= H && ox0 >= 1 && ox0 + 17 <= W) {   // the whole halo is inside the image: no predicates
#pragma unroll
      for (int hy = 0; hy < C64_HR; ++hy)
#pragma unroll
        for (int g = 0; g < 2; ++g) pre[hy * 2 + g] = C64_LD(org + (long long)hy * W * 64 + lane * 8 + g * 512);
#pragma unroll
      for (int e = 0; e < C64_NE; ++e) {
        pre[2 * C64_HR + e] = make_uint4(0, 0, 0, 0);
        if (4 * e + er < C64_HR) pre[2 * C64_HR + e] = C64_LD(org + (long long)e * 4 * W * 64 + geoff);
      }
      prem = (1u << C64_NCH) - 1u;
      return;
    }
    // (both bounds on both column groups: a ragged last patch column — W % 16 in 1..6 — ends inside group 0)
    const bool cok0 = (unsigned)(ox0 - 1 + lx) < (unsigned)W, cok1 = (unsigned)(ox0 + 7 + lx) < (unsigned)W;
    unsigned m = 0;
#pragma unroll
    for (int hy = 0; hy < C64_HR; ++hy) {
      const bool rok = (unsigned)(oy0 - 1 + hy) < (unsigned)H;
      const T* rp = org + (long long)hy * W * 64 + lane * 8;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const bool ok = rok && (g ? cok1 : cok0);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = C64_LD(rp + g * 512);
        pre[hy * 2 + g] = v;
        if (XF != 0 && ok) m |= 1u << (hy * 2 + g);
      }
    }
#pragma unroll
    for (int e = 0; e < C64_NE; ++e) {
      const int hy = 4 * e + er;
      const bool ok = hy < C64_HR && (unsigned)(oy0 - 1 + hy) < (unsigned)H && ox0 + 15 + ej < W;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = C64_LD(org + (long long)e * 4 * W * 64 + geoff);
      pre[2 * C64_HR + e] = v;
      if (XF != 0 && ok) m |= 1u << (2 * C64_HR + e);
    }
    prem = m;
  };
  // (the transform runs HERE, a whole patch after the loads were issued: at load time it would wait for them)
  auto store_halo = [&]() {
#pragma unroll
    for (int hy = 0; hy < C64_HR; ++hy)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        *reinterpret_cast<uint4*>(hb + sb0 + hy * (C64_HP * 128) + g * 1024) = xform(pre[hy * 2 + g], (prem >> (hy * 2 + g)) & 1u);
#pragma unroll
    for (int e = 0; e < C64_NE; ++e)
      if (4 * e + er < C64_HR)
        *reinterpret_cast<uint4*>(hb + sbe + e * 4 * C64_HP * 128) = xform(pre[2 * C64_HR + e], (prem >> (2 * C64_HR + e)) & 1u);
  };

  // The output stores of a patch are DEFERRED to the next iteration, in front of the request for the halo after next.  With
  // the stores issued right after the MFMAs they were younger than the halo loads in flight, and the waits hipcc derives
  // for the LDS refill (it counts the loads only) then ran down to vmcnt(0): every refill waited for the previous patch's
  // stores to be acknowledged — a microsecond with nothing else to run on the SIMD.  Deferred, nothing younger than the
  // halo loads is in flight when they are waited for, and the stores had a whole K loop to retire.
  uint4 outp[2 * C64_PR];   // the patch's packed bf16 rows, [r][q]
  T* out_ptr = nullptr;     // where they go (per lane), nullptr: nothing pending
  unsigned out_ok = 0;      // bit r: row r of the patch is inside the image for this lane
  auto flush_out = [&]() {
    if (out_ptr) {
#pragma unroll
      for (int r = 0; r < C64_PR; ++r) {
        if (!((out_ok >> r) & 1u)) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) C64_ST(out_ptr + (long long)r * W * 64 + q * 32, outp[r * 2 + q]);
      }
    }
  };

  int p = gw;
  if (p < P) load_halo(p);
  while (p < P) {
    int n, oy0, ox0;
    patch_origin(p, n, oy0, ox0);
    store_halo();
    flush_out();
    const int pn = p + gstride;
    if (pn < P) load_halo(pn);     // in flight under this patch's MFMAs

    f32x4 acc[C64_PR][4];               // [pixel row r][channel tile t]
#pragma unroll
    for (int r = 0; r < C64_PR; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // K loop, 18 steps (32-channel chunk major, tap minor).  Fragments are double-buffered by hand: the reads of step
      // k + 1 are issued before the 16 MFMAs of step k (one wave has at most one partner on its SIMD: LDS latency is not
      // hidden by occupancy here)
      frag_t wf[2][4], pf[2][C64_PR];
      auto ldfr = [&](auto stc, auto bc) {
        constexpr int st = decltype(stc)::value, b = decltype(bc)::value;
        constexpr int s = st / 9, tap = st % 9, kh = tap / 3, kw = tap % 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[b][t] = *reinterpret_cast<const frag_t*>(wbs[s] + tap * 8192 + t * 2048);
#pragma unroll
        for (int r = 0; r < C64_PR; ++r) pf[b][r] = *reinterpret_cast<const frag_t*>(hb + (aoff[kw] ^ (s ? 64 : 0)) + (r + kh) * (C64_HP * 128));
      };
      ldfr(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
      static_for<18>([&](auto stc) {
        constexpr int st = decltype(stc)::value, b = st & 1;
        if constexpr (st + 1 < 18) ldfr(std::integral_constant<int, st + 1>(), std::integral_constant<int, b ^ 1>());
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < C64_PR; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[r][t] = Mma<T>::run(wf[b][t], pf[b][r], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- epilogue, from the registers: lane (frow, fg) holds channels 32 q + 8 fg .. + 7 of pixel (oy0 + r, ox0 + frow) ----
    const bool full = oy0 + C64_PR <= H && ox0 + 16 <= W;
    const bool colok = ox0 + frow < W;
    const long long yoff = (((long long)n * H + oy0) * W + ox0 + frow) * 64 + fg * 8;
    T* yp = y + yoff;
    if constexpr (EMODE == 0) {
      if (a.stats) {
#pragma unroll
        for (int r = 0; r < C64_PR; ++r) {
          const bool ok = full || (colok && oy0 + r < H);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = ok ? acc[r][t][e] : 0.f;
              s1[(t >> 1) * 8 + (t & 1) * 4 + e] += v;
              s2[(t >> 1) * 8 + (t & 1) * 4 + e] += v * v;
            }
        }
      }
      out_ok = 0;
#pragma unroll
      for (int r = 0; r < C64_PR; ++r) {
        if (full || (colok && oy0 + r < H)) out_ok |= 1u << r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 u = acc[r][2 * q], v = acc[r][2 * q + 1];
          outp[r * 2 + q] = make_uint4(pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3]), pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
      }
      out_ptr = yp;
    } else {
      out_ok = 0;
#pragma unroll
      for (int r = 0; r < C64_PR; ++r)
        if (full || (colok && oy0 + r < H)) out_ok |= 1u << r;
      out_ptr = yp;
      const bool has_mask = a.mask != nullptr, from_y = a.ep1 && a.ep2;
      // (requesting these operands before the MFMAs was measured: the registers they hold through the K loop cost more
      //  in spills than the hidden latency gains — 1.85 -> 2.15 ms)
      const T* __restrict__ etp = reinterpret_cast<const T*>(a.et) + yoff;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        Vec16<T> vy[C64_PR];
        unsigned mb8[C64_PR];
#pragma unroll
        for (int r = 0; r < C64_PR; ++r) {
          vy[r].zero();
          mb8[r] = 0;
          if (full || (colok && oy0 + r < H)) {
            const long long ro = (long long)r * W * 64 + q * 32;
            vy[r].load(etp + ro);
            if (a.mask && a.mask_bits) mb8[r] = reinterpret_cast<const unsigned char*>(a.mask)[(yoff + ro) >> 3];
          }
        }
        float q0[8], q1[8], q2[8];
        {
          const float* cs = ctab + 128 + q * 32 + fg * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            q0[e] = cs[e];
            q1[e] = cs[64 + e];
            q2[e] = cs[128 + e];
          }
        }
        Vec16<T> vm[C64_PR];
        bool ok[C64_PR];
#pragma unroll
        for (int r = 0; r < C64_PR; ++r) {
          ok[r] = full || (colok && oy0 + r < H);
          vm[r].zero();
          if (ok[r] && has_mask && !a.mask_bits) vm[r].load(reinterpret_cast<const T*>(a.mask) + yoff + (long long)r * W * 64 + q * 32);
        }
#pragma unroll
        for (int r = 0; r < C64_PR; ++r) {
          if (!ok[r]) continue;
          const f32x4 u = acc[r][2 * q], v = acc[r][2 * q + 1];
          Vec16<T> o;
          o.raw = make_uint4(pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3]), pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          float fv[8], fy[8];
          o.get(fv);     // the bf16 values of the unmasked gradient, as the C tile of the halo kernel holds them
          vy[r].get(fy);
          if (has_mask) {
            if (a.mask_bits) {
#pragma unroll
              for (int e = 0; e < 8; ++e) fv[e] = ((mb8[r] >> e) & 1u) ? fv[e] : 0.f;
            } else {
              float fm[8];
              vm[r].get(fm);
#pragma unroll
              for (int e = 0; e < 8; ++e) fv[e] = fm[e] > 0.f ? fv[e] : 0.f;
            }
          } else if (from_y) {   // the unit below's ReLU output is positive exactly where y*scale + shift is
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = (fy[e] * q1[e] + q2[e]) > 0.f ? fv[e] : 0.f;
          }
          o.set(fv);     // (exact: the values are bf16 or zero)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[q * 8 + e] += fv[e];
            s2[q * 8 + e] += fv[e] * (fy[e] - q0[e]);
          }
          outp[r * 2 + q] = o.raw;
        }
      }
    }
    p = pn;
  }
  flush_out();

  // ---- one slab row per wave: sums over the 16 pixel lanes of each channel group ----
  const int srow = blockIdx.x * C64_NW + wid;
  if (a.stats && srow < gstride) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float u = s1[i], v = s2[i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        u += __shfl_xor(u, o);
        v += __shfl_xor(v, o);
      }
      if (frow == 0) {
        const int co = 32 * (i >> 3) + 8 * fg + (i & 7);
        a.stats[((long long)srow * 2 + 0) * 64 + co] = u;
        a.stats[((long long)srow * 2 + 1) * 64 + co] = v;
      }
    }
  }
}

// Workgroups launched for this geometry (slab rows = one per wave): one per CU, fewer for small inputs.
int maai_conv_c64_rows(const ConvArgs& a) {
  const long long P = (long long)a.N * ((a.IW + 15) / 16) * ((a.IH + 6) / 7);   // (16 x 7 patches: the coarser of the two sizes)
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long wgs = (P + C64_NW - 1) / C64_NW;
  if (wgs > cus) wgs = cus;
  return (int)(wgs * C64_NW);
}

bool maai_conv_c64_supported(const ConvArgs& a, int dtype) {
  if (dtype != MAAI_BF16 || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad_h != 1 || a.pad_w != 1 || a.Cin != 64 || a.Cout != 64) return false;
  if (a.OHg != a.IH || a.OWg != a.IW || a.OH != a.OHg || a.OW != a.OWg || a.ostr != 1 || a.ooh || a.oow) return false;
  if (a.xb || a.a2 || a.pre_x || a.accumulate) return false;
  if ((long long)a.N * ((a.IW + 15) / 16) * ((a.IH + 5) / 6) >= (1ll << 30)) return false;   // (32-bit patch arithmetic)
  if (a.emode == MAAI_EPI_STORE) return !a.mask;
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) return a.et && !a.xs;
  return false;
}

template <int XF, int EMODE>
static int launch_c64(const ConvArgs& a, hipStream_t st) {
  constexpr int lds = C64_WB + C64_NW * c64_hb(c64_pr(EMODE)) + 320 * 4;   // weights | one halo image per wave | coefficient tables
  static int attr[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&conv_c64_kernel<XF, EMODE>), lds, attr);
  const int grid = maai_conv_c64_rows(a) / C64_NW;
  MAAI_NOTE_KERNEL(conv_c64_kernel<XF, EMODE>);
  hipLaunchKernelGGL((conv_c64_kernel<XF, EMODE>), dim3((unsigned)grid), dim3(64 * C64_NW), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

int maai_conv_c64_launch(const ConvArgs& a, hipStream_t st) {
  if (a.emode == MAAI_EPI_DGRAD_REDUCE) return launch_c64<0, 6>(a, st);
  return a.xs ? launch_c64<1, 0>(a, st) : launch_c64<0, 0>(a, st);
}
