// Gram matrix and column sums of an activation tensor: Gram = x^T x [C, C] (fp32), sx = colsum(x) [C] (fp64), x = [M, C] bf16 —
// the two pixel reductions over a unit's INPUT that the folded BatchNorm backward needs beside G1 = g^T x (csrc/fold.hip,
// engine._FOLD; conv3 + bn3 of the reference's Bottleneck, resnet.py:118-119): dW = k1*G1 - k2 (x) sx - k3*(W Gram).
// x may be a normalise-on-load activation (xs != null): x = act(y*xs + xt) formed on the way into LDS with the arithmetic of
// maai_bn_act_fwd — layer 1's conv3 input is never stored.
//
// Persistent workgroups walk 64-pixel tiles: the tile goes global -> registers (prefetched one tile ahead) -> LDS as C/64
// pixel-major sub-images of [64 pixels][64 channels] (128-byte rows, the swizzle of conv_bwd3.hip), and both MFMA operands
// of x^T x are read from that ONE image with transposed reads (ds_read_b64_tr_b16: the contraction index is the row).  A
// workgroup of class y owns rows [y*CR, (y+1)*CR) of the Gram matrix (CR = min(C, 128)): wave w owns CR/4 of them against all C
// columns, accumulators in registers for the whole launch, one fp32 atomic flush at the end (a backward-only quantity: like
// every weight gradient of this library it is not bit-reproducible from run to run).
#include "conv_igemm.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_g;

struct GramArgs {
  const void* x;
  const float* xs;
  const float* xt;
  int x_relu;
  float* gram;   // [C][C], += (zeroed by the caller)
  double* sx;    // [C], += (zeroed by the caller)
  double* npos;  // [C], += the number of rows with x > 0 (nullable)
  float* pgram;  // deterministic mode (nullable): partial Gram matrices [gridDim.x][C][C] written, not accumulated — and
  float* psx;    //   partial column sums [gridDim.x][C]; summed by the caller in a fixed order (maai_reduce_partials)
  long long M;
  int ntiles;
};

template <int C>
struct GramPT {
  static constexpr int value = C <= 64 ? 4 : (C <= 128 ? 2 : 1);
};
static long long gram_ntiles(long long M, int C) {
  const int px = 64 * (C <= 64 ? 4 : (C <= 128 ? 2 : 1));
  return (M + px - 1) / px;
}

template <int C, bool XF>
__global__ __launch_bounds__(256, C <= 128 ? 2 : 1) void gram_kernel(GramArgs a) {
  typedef bf16_t T;
  constexpr int CR = C < 128 ? C : 128;          // Gram rows of this workgroup class
  constexpr int TR = CR / 64;                    // 16-row blocks per wave
  constexpr int TC = C / 16;                     // 16-column blocks
  constexpr int RB = 128, SUB = 64 * RB;         // C / 64 sub-images of [64 px][64 ch]
  constexpr int CPR = C / 8;                     // 16-byte chunks per pixel
  constexpr int NLD = CPR / 4;                   // chunks per thread per 64-pixel slice (64 * CPR / 256)
  constexpr int PT = GramPT<C>::value;           // 64-pixel slices per tile: a narrow image gets a longer tile (bytes in flight, work per barrier pair)
  constexpr int NSUB = C / 64;                   // sub-images per slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [256 / CPR][C] partial column sums: reuses the image after the last tile (8 KB <= the image)

  const int tid = threadIdx.x, lane = tid & 63;
  const int widu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gl = lane >> 4;
  const int cls = blockIdx.y;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);

  // ---- this thread's chunk column: channels cir*8 .. +8 of pixels p0 + i * (256 / CPR) ----
  const int cir = tid % CPR, p0 = tid / CPR;
  constexpr int PSTEP = 256 / CPR;
  float qs[8], qt[8], csum[8], cpos[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    qs[e] = XF ? a.xs[cir * 8 + e] : 1.f;
    qt[e] = XF ? a.xt[cir * 8 + e] : 0.f;
    csum[e] = 0.f;
    cpos[e] = 0.f;
  }
  const int sub = cir >> 3, chunk = cir & 7;

  f32x4 acc[TR][TC];
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing inside a [64 px][64 ch] sub-image (conv_bwd3.hip): 16-channel block b of the sub-image, K-step s
  const int q = li >> 2, p = li & 3, r0 = 4 * gl + q;
  int offb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) offb[b] = r0 * RB + (((2 * b + (p >> 1)) ^ (((r0 >> 1) & 3) << 1)) << 4) + ((p & 1) << 3);

  uint4 pre[PT * NLD];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        long long m = ((long long)t * PT + pt) * 64 + p0 + i * PSTEP;
        if (m >= a.M) m = a.M - 1;
        pre[pt * NLD + i] = ld16_nt(x + m * C + cir * 8);
      }
  };
  auto store_tile = [&](int t) {
#pragma unroll
    for (int pi = 0; pi < PT * NLD; ++pi) {
      const int pt = pi / NLD, i = pi % NLD;
      const int px = p0 + i * PSTEP;
      Vec16<T> v, w2;
      v.raw = pre[pi];
      if constexpr (XF) XfMath<T>::template run<false>(v, w2, qs, qt, nullptr, nullptr, a.x_relu, false);
      if (((long long)t * PT + pt) * 64 + px >= a.M) v.zero();
      if (cls == 0) {
        float f[8];
        v.get(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          csum[e] += f[e];
          cpos[e] += f[e] > 0.f ? 1.f : 0.f;   // (exact in fp32: a thread sees < 2^24 rows)
        }
      }
      *reinterpret_cast<uint4*>(smem + (pt * NSUB + sub) * SUB + px * RB + ((chunk ^ (((px >> 1) & 3) << 1)) << 4)) = v.raw;
    }
  };

  int t = blockIdx.x;
  if (t < a.ntiles) load_tile(t);
  for (; t < a.ntiles; t += gridDim.x) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // everyone has finished reading the previous tile's image
    store_tile(t);
    if (t + (int)gridDim.x < a.ntiles) load_tile(t + gridDim.x);   // in flight under the multiply phase
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the image is complete
#pragma unroll
    for (int ps = 0; ps < 2 * PT; ++ps) {
      const int s = ps & 1;
      const char* img = smem + (ps >> 1) * NSUB * SUB;
      bf16x8 af[TR];
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int ch16 = cls * (CR / 16) + widu * TR + r;     // 16-channel block of the Gram row block
        const char* base = img + (ch16 >> 2) * SUB + s * 32 * RB + offb[ch16 & 3];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_g*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_g*)(base + 16 * RB));
        af[r] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        const char* base = img + (c >> 2) * SUB + s * 32 * RB + offb[c & 3];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_g*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_g*)(base + 16 * RB));
        const bf16x8 bfr = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int r = 0; r < TR; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r], bfr, acc[r][c], 0, 0, 0);
      }
    }
  }
  // ---- flush: C layout row = 4*gl + e, column = li ----
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = cls * CR + (widu * TR + r) * 16 + 4 * gl + e;
        if (a.pgram) a.pgram[((long long)blockIdx.x * C + row) * C + c * 16 + li] = acc[r][c][e];
        else atomicAdd(a.gram + (long long)row * C + c * 16 + li, acc[r][c][e]);
      }
  if (cls == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the image is dead: its memory holds the partial column sums now
#pragma unroll
    for (int e = 0; e < 8; ++e) red[p0 * C + cir * 8 + e] = csum[e];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      double s = 0.0;
      for (int k = 0; k < PSTEP; ++k) s += (double)red[k * C + c];
      if (a.psx) a.psx[(long long)blockIdx.x * C + c] = (float)s;
      else atomicAdd(a.sx + c, s);
    }
    if (a.npos) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[p0 * C + cir * 8 + e] = cpos[e];
      __syncthreads();
      for (int c = tid; c < C; c += 256) {
        double s = 0.0;
        for (int k = 0; k < PSTEP; ++k) s += (double)red[k * C + c];
        atomicAdd(a.npos + c, s);
      }
    }
  }
}

template <int C>
static int launch_gram(const GramArgs& a, hipStream_t st) {
  constexpr int lds = GramPT<C>::value * (C / 64) * 64 * 128;   // (>= the column-sum scratch: 256/CPR rows of C floats = 8 KB)
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  constexpr int ncls = C / (C < 128 ? C : 128);
  long long gx = (long long)cus * (C <= 128 ? 2 : 1) / ncls;
  if (gx < 1) gx = 1;
  if (gx > a.ntiles) gx = a.ntiles;
  static int attr[2][64] = {{0}};
  if (a.xs) {
    maai_ensure_lds(reinterpret_cast<const void*>(&gram_kernel<C, true>), lds, attr[0]);
    MAAI_NOTE_KERNEL(gram_kernel<C, true>);
    hipLaunchKernelGGL((gram_kernel<C, true>), dim3((unsigned)gx, ncls), dim3(256), lds, st, a);
  } else {
    maai_ensure_lds(reinterpret_cast<const void*>(&gram_kernel<C, false>), lds, attr[1]);
    MAAI_NOTE_KERNEL(gram_kernel<C, false>);
    hipLaunchKernelGGL((gram_kernel<C, false>), dim3((unsigned)gx, ncls), dim3(256), lds, st, a);
  }
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_gram_partial_rows(long long M, int C) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int ncls = C / (C < 128 ? C : 128);
  long long gx = (long long)cus * (C <= 128 ? 2 : 1) / ncls;
  const long long ntiles = gram_ntiles(M, C);
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  return (int)gx;
}

// Deterministic form: partial Gram matrices pgram [maai_gram_partial_rows(M, C)][C][C] and partial column sums psx [rows][C]
// (fp32, one row per workgroup column, every element written) — no atomics: the caller sums the rows in a fixed order
// (maai_reduce_partials), so the result is bit-reproducible and may feed FORWARD statistics (maai_fold_stats).
extern "C" int maai_gram_partials(const void* x, long long M, int C, const float* xs, const float* xt, int x_relu, float* pgram,
                                  float* psx, void* stream) {
  MAAI_CHECK_ARG(x && pgram && psx && M > 0 && (xs == nullptr) == (xt == nullptr), "gram_partials: bad arguments");
  MAAI_CHECK_ARG(M < (1ll << 31), "gram_partials: pixel count must fit 31 bits");
  GramArgs a;
  a.x = x; a.xs = xs; a.xt = xt; a.x_relu = x_relu; a.gram = nullptr; a.sx = nullptr; a.npos = nullptr; a.pgram = pgram; a.psx = psx;
  a.M = M; a.ntiles = (int)gram_ntiles(M, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 64: return launch_gram<64>(a, st);
    case 128: return launch_gram<128>(a, st);
    case 256: return launch_gram<256>(a, st);
    case 512: return launch_gram<512>(a, st);
    default: break;
  }
  maai_set_error("gram: built for 64, 128, 256 and 512 channels");
  return MAAI_ERR_UNSUPPORTED;
}

// x [M][C] bf16 (C = 64, 128, 256 or 512), optionally normalised on load (xs, xt [C], x_relu); gram [C][C] fp32, sx [C] fp64
// and npos [C] fp64 (nullable: rows with x > 0 per channel) are ACCUMULATED into (zero them first).
extern "C" int maai_gram(const void* x, long long M, int C, const float* xs, const float* xt, int x_relu, float* gram, double* sx,
                         double* npos, void* stream) {
  MAAI_CHECK_ARG(x && gram && sx && M > 0 && (xs == nullptr) == (xt == nullptr), "gram: bad arguments");
  MAAI_CHECK_ARG(M < (1ll << 31), "gram: pixel count must fit 31 bits");
  GramArgs a;
  a.x = x; a.xs = xs; a.xt = xt; a.x_relu = x_relu; a.gram = gram; a.sx = sx; a.npos = npos; a.pgram = nullptr; a.psx = nullptr; a.M = M; a.ntiles = (int)gram_ntiles(M, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (C) {
    case 64: return launch_gram<64>(a, st);
    case 128: return launch_gram<128>(a, st);
    case 256: return launch_gram<256>(a, st);
    case 512: return launch_gram<512>(a, st);
    default: break;
  }
  maai_set_error("gram: built for 64, 128, 256 and 512 channels");
  return MAAI_ERR_UNSUPPORTED;
}
