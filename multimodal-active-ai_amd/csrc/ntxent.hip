// NT-Xent contrastive loss (Objective.py:17-81,123-125), fp32 throughout.
//
// The reference builds four [B,N] logit matrices, masks, concatenates and runs
// log_softmax over rows of length 2N.  Here every 16 x 16 similarity tile is
// produced on the matrix cores with the exact-fp32 v_mfma_f32_16x16x4_f32 and
// consumed in registers: online log-sum-exp in the forward pass, softmax minus
// one-hot re-materialised from the saved row LSE in the backward pass, whose
// tile is fed straight back into the matrix cores as the A operand of the
// gradient product (cdna_hip_programming.md §3 "An accumulator tile as the
// next MFMA's operand").  Only logits_ab (a return value of the reference
// API) is ever written to HBM.
//
// Orientation: a tile is S^T — MFMA rows = 16 "source" vectors (columns of the
// reference's logits), MFMA cols = 16 "target" vectors (softmax rows in the
// forward pass).  C layout: source = 4*(lane>>4)+reg, target = lane&15, so each
// lane owns ONE target and the row state (max, sum) is one scalar pair per lane.
#include "common.h"
#include "maai_internal.h"

#define NTX_LARGE 1e9f
#define ST(stream) reinterpret_cast<hipStream_t>(stream)

// ---------------------------------------------------------------------------
// z = h / max(||h||, 1e-12)   (F.normalize, Objective.py:42-43)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ h, float* __restrict__ z,
                                                        float* __restrict__ inv_norm, int B, int d, int normalize) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float v = h[(long long)row * d + k];
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float den = normalize ? fmaxf(sqrtf(ss), 1e-12f) : 1.f;
  for (int k = lane; k < d; k += 64) z[(long long)row * d + k] = h[(long long)row * d + k] / den;
  if (lane == 0 && inv_norm) inv_norm[row] = 1.f / den;
}

extern "C" int maai_ntxent_normalize(const float* h, float* z, float* inv_norm, int B, int d, int normalize, void* stream) {
  MAAI_CHECK_ARG(h && z && B > 0 && d > 0, "ntxent_normalize: bad arguments");
  hipLaunchKernelGGL(normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(stream), h, z, inv_norm, B, d, normalize);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// dh = (dz - z (z.dz)) * inv_norm
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dz,
                                                            const float* __restrict__ inv_norm, float* __restrict__ dh, int B,
                                                            int d, int normalize) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float dot = 0.f;
  if (normalize) {
    for (int k = lane; k < d; k += 64) dot += z[(long long)row * d + k] * dz[(long long)row * d + k];
    dot = wave_sum(dot);
  }
  const float inv = normalize ? inv_norm[row] : 1.f;
  for (int k = lane; k < d; k += 64) {
    const long long o = (long long)row * d + k;
    dh[o] = normalize ? (dz[o] - z[o] * dot) * inv : dz[o];
  }
}

extern "C" int maai_ntxent_normalize_bwd(const float* z, const float* dz, const float* inv_norm, float* dh, int B, int d,
                                         int normalize, void* stream) {
  MAAI_CHECK_ARG(z && dz && dh && B > 0 && d > 0 && (!normalize || inv_norm), "ntxent_normalize_bwd: bad arguments");
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(stream), z, dz, inv_norm, dh, B, d, normalize);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// S^T tile on the matrix cores: acc[r] = sum_k Q[q0+4g+r][k] * P[t][k]
// P: 16 target rows staged in LDS with pitch d+4; Q: global rows (L2 resident).
// ---------------------------------------------------------------------------
__device__ __forceinline__ f32x4 st_tile(const float* __restrict__ Q, int q0, int qcnt, const float* Pl, int d, int lane) {
  const int r = lane & 15, g = lane >> 4;
  int qr = q0 + r;
  if (qr > qcnt - 1) qr = qcnt - 1;
  const float* qp = Q + (long long)qr * d + g * 4;
  const float* pp = Pl + r * (d + 4) + g * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < d; kb += 16) {
    const float4 a = *reinterpret_cast<const float4*>(qp + kb);
    const float4 b = *reinterpret_cast<const float4*>(pp + kb);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ void stage_targets(const float* __restrict__ P, int t0, int tcnt, int d, float* Pl) {
  // 16 rows x d floats -> LDS pitch d+4; rows past tcnt replicate the last row (results discarded)
  const int nv = 16 * (d / 4);
  for (int i = threadIdx.x; i < nv; i += 256) {
    const int r = i / (d / 4), c = (i - r * (d / 4)) * 4;
    int tr = t0 + r;
    if (tr > tcnt - 1) tr = tcnt - 1;
    *reinterpret_cast<float4*>(Pl + r * (d + 4) + c) = *reinterpret_cast<const float4*>(P + (long long)tr * d + c);
  }
}

// ---------------------------------------------------------------------------
// forward: per softmax row (type a: z1_i vs [Z2 | Z1 masked]; type b: z2_i vs [Z1 | Z2 masked])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ntxent_fwd_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                         const float* __restrict__ Z1, const float* __restrict__ Z2,
                                                         float* __restrict__ logits_ab, float* __restrict__ lse, int B, int N,
                                                         int d, float inv_tau, int row_offset) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Pl = sm;                       // [16][d+4]
  float* red = sm + 16 * (d + 4);       // [4 waves][16][3]
  const int type = blockIdx.y, t0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t = lane & 15, g = lane >> 4;
  const float* P = type == 0 ? z1 : z2;
  stage_targets(P, t0, B, d, Pl);
  __syncthreads();
  const int tiles_half = (N + 15) / 16;
  const int row = t0 + t;                 // local softmax row of this lane
  const int gpos = row_offset + row;      // its column in the gathered set
  float m = -INFINITY, s = 0.f, pos = 0.f;
  for (int tile = wid; tile < 2 * tiles_half; tile += 4) {
    const int half = tile >= tiles_half;
    const int q0 = (tile - half * tiles_half) * 16;
    const float* Q = (type == 0) ? (half ? Z1 : Z2) : (half ? Z2 : Z1);
    const f32x4 acc = st_tile(Q, q0, N, Pl, d, lane);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      float x = acc[r] * inv_tau;
      if (half && q == gpos) x -= NTX_LARGE;        // logits_aa/bb - masks*LARGE_NUM (Objective.py:68,71)
      if (!half && q == gpos) pos += x;             // the one-hot label column (Objective.py:55-57)
      v[r] = q < N ? x : -INFINITY;
    }
    if (type == 0 && !half && row < B) {
      float* dst = logits_ab + (long long)row * N + q0 + 4 * g;
      if (q0 + 4 * g + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (q0 + 4 * g + r < N) dst[r] = v[r];
      }
    }
    const float tm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (tm > -INFINITY) {
      const float nm = fmaxf(m, tm);
      s = s * expf(m - nm) + expf(v[0] - nm) + expf(v[1] - nm) + expf(v[2] - nm) + expf(v[3] - nm);
      m = nm;
    }
  }
  // merge the 4 lane groups that share a target (lanes t, t+16, t+32, t+48)
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float om = __shfl_xor(m, o), os = __shfl_xor(s, o), op = __shfl_xor(pos, o);
    const float nm = fmaxf(m, om);
    const float sa = (m > -INFINITY) ? s * expf(m - nm) : 0.f;
    const float sb = (om > -INFINITY) ? os * expf(om - nm) : 0.f;
    s = sa + sb;
    m = nm;
    pos += op;
  }
  if (lane < 16) {
    red[(wid * 16 + t) * 3 + 0] = m;
    red[(wid * 16 + t) * 3 + 1] = s;
    red[(wid * 16 + t) * 3 + 2] = pos;
  }
  __syncthreads();
  if (threadIdx.x < 16 && t0 + threadIdx.x < B) {
    float M = -INFINITY, S = 0.f, Pp = 0.f;
    for (int w = 0; w < 4; ++w) {
      const float om = red[(w * 16 + threadIdx.x) * 3], os = red[(w * 16 + threadIdx.x) * 3 + 1];
      Pp += red[(w * 16 + threadIdx.x) * 3 + 2];
      if (om > -INFINITY) {
        const float nm = fmaxf(M, om);
        S = (M > -INFINITY ? S * expf(M - nm) : 0.f) + os * expf(om - nm);
        M = nm;
      }
    }
    const float l = M + logf(S);
    const int r = t0 + threadIdx.x;
    lse[type * B + r] = l;
    lse[(2 + type) * B + r] = l - Pp;  // -log softmax at the label column
  }
}

__global__ __launch_bounds__(256) void ntxent_loss_sum_kernel(const float* __restrict__ rowloss, int n, float inv_b, float* loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += rowloss[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) * inv_b;
}

extern "C" int maai_ntxent_fwd(const float* z1, const float* z2, const float* Z1, const float* Z2, float* loss,
                               float* logits_ab, float* lse, int B, int N, int d, float temperature, int row_offset,
                               void* stream) {
  MAAI_CHECK_ARG(z1 && z2 && Z1 && Z2 && loss && logits_ab && lse, "ntxent_fwd: null pointer");
  MAAI_CHECK_ARG(B > 0 && N >= B && d > 0 && d % 16 == 0 && d <= 496, "ntxent_fwd: need N >= B, d % 16 == 0, d <= 496 (the backward pass holds 16 x (d + 4) + 64 x d floats in the 160 KiB of LDS)");
  MAAI_CHECK_ARG(row_offset >= 0 && row_offset + B <= N && temperature > 0.f, "ntxent_fwd: bad row_offset / temperature");
  const size_t lds = (16 * (d + 4) + 4 * 16 * 3) * sizeof(float);
  hipLaunchKernelGGL(ntxent_fwd_kernel, dim3((B + 15) / 16, 2), dim3(256), lds, ST(stream), z1, z2, Z1, Z2, logits_ab, lse, B, N,
                     d, 1.0f / temperature, row_offset);
  MAAI_CHECK_LAUNCH();
  hipLaunchKernelGGL(ntxent_loss_sum_kernel, dim3(1), dim3(256), 0, ST(stream), lse + 2 * B, 2 * B, 1.0f / (float)B, loss);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// ---------------------------------------------------------------------------
// backward.  out[t][:] (+)= sum over segments, q:  G[q][t] * Q[q][:] / tau
//   G = (exp(S - LSE[row]) - [label]) * gloss / B,  row = target (left-operand
//   gradient) or source (right-operand gradient).
// ---------------------------------------------------------------------------
struct NtxSeg {
  const float* Q;      // source vectors [cnt][d]
  const float* lse_q;  // LSE indexed by source (right-operand pass) or nullptr (row = target)
  int cnt;
  int toff;            // relation tested: q == t + toff
  int rule;            // 0 none, 1 label column (positive), 2 self mask
};
struct NtxBwdArgs {
  const float* P;      // target vectors [tcnt][d]
  const float* lse_t;  // LSE indexed by target
  float* out;          // [tcnt][d]
  const float* gloss;  // scalar upstream gradient (device)
  int tcnt, d, nseg;
  float inv_tau, inv_b;
  NtxSeg seg[4];
};

__global__ __launch_bounds__(256) void ntxent_bwd_kernel(NtxBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = a.d;
  float* Pl = sm;                   // [16][d+4]
  float* red = sm + 16 * (d + 4);   // [4 waves][16][d]
  const int t0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t = lane & 15, g = lane >> 4;
  stage_targets(a.P, t0, a.tcnt, d, Pl);
  __syncthreads();
  const int NT = d / 16;  // output tiles along d (<= 32)
  f32x4 acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int trow = t0 + t;
  const float lse_t = (a.lse_t && trow < a.tcnt) ? a.lse_t[trow] : 0.f;
  const float scale = a.inv_b * (a.gloss ? *a.gloss : 1.f);
  int tile_base = 0;
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const NtxSeg sg = a.seg[sgi];
    const int ntile = (sg.cnt + 15) / 16;
    for (int tile = wid - (tile_base & 3); tile < ntile; tile += 4) {
      if (tile < 0) continue;
      const int q0 = tile * 16;
      const f32x4 s = st_tile(sg.Q, q0, sg.cnt, Pl, d, lane);
      float G[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        const bool rel = (q == trow + sg.toff);
        float x = s[r] * a.inv_tau;
        if (sg.rule == 2 && rel) x -= NTX_LARGE;
        const float l = sg.lse_q ? sg.lse_q[q < sg.cnt ? q : sg.cnt - 1] : lse_t;
        float p = expf(x - l);
        if (sg.rule == 1 && rel) p -= 1.f;
        G[r] = (q < sg.cnt && trow < a.tcnt) ? p * scale : 0.f;
      }
      // out[t][n] += sum_q G[q][t] * Q[q][n] : A[t][k=g] = G[4g+r][t] (reg r), B[k=g][n] = Q[q0+4g+r][n]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int qr = q0 + 4 * g + r;
        if (qr > sg.cnt - 1) qr = sg.cnt - 1;
        const float* qrow = sg.Q + (long long)qr * d + t;
#pragma unroll
        for (int n = 0; n < 32; ++n)
          if (n < NT) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(G[r], qrow[n * 16], acc[n], 0, 0, 0);
      }
    }
    tile_base += ntile;
  }
  // acc[n][r]: row (target) = 4g + r, col = n*16 + t  -> combine the four waves through LDS
#pragma unroll
  for (int n = 0; n < 32; ++n)
    if (n < NT)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid * 16 + 4 * g + r) * d + n * 16 + t] = acc[n][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * d; i += 256) {
    const int r = i / d, c = i - r * d;
    if (t0 + r < a.tcnt) {
      const float v = (red[i] + red[16 * d + i] + red[32 * d + i] + red[48 * d + i]) * a.inv_tau;
      a.out[(long long)(t0 + r) * d + c] = v;
    }
  }
}

static int launch_bwd(const NtxBwdArgs& a, hipStream_t st) {
  const size_t lds = (16 * (a.d + 4) + 4 * 16 * a.d) * sizeof(float);
  static int attr[64] = {0};
  maai_ensure_lds(reinterpret_cast<const void*>(&ntxent_bwd_kernel), (int)lds, attr);
  hipLaunchKernelGGL(ntxent_bwd_kernel, dim3((a.tcnt + 15) / 16), dim3(256), lds, st, a);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

extern "C" int maai_ntxent_bwd(const float* z1, const float* z2, const float* Z1, const float* Z2, const float* lse,
                               const float* gloss, float* dz1, float* dz2, int B, int N, int d, float temperature,
                               int row_offset, int local_in_gathered, void* stream) {
  MAAI_CHECK_ARG(z1 && z2 && Z1 && Z2 && lse && dz2, "ntxent_bwd: null pointer");
  MAAI_CHECK_ARG(B > 0 && N >= B && d > 0 && d % 16 == 0 && d <= 496, "ntxent_bwd: need N >= B, d % 16 == 0, d <= 496 (16 x (d + 4) + 64 x d floats must fit the 160 KiB of LDS)");
  MAAI_CHECK_ARG(row_offset >= 0 && row_offset + B <= N && temperature > 0.f, "ntxent_bwd: bad row_offset / temperature");
  hipStream_t st = ST(stream);
  const float* lse_a = lse;
  const float* lse_b = lse + B;
  NtxBwdArgs a;
  a.gloss = gloss; a.tcnt = B; a.d = d; a.inv_tau = 1.f / temperature; a.inv_b = 1.f / (float)B;
  // ---- dz2 ----
  a.P = z2; a.lse_t = lse_b; a.out = dz2; a.nseg = 0;
  // left operand of ba (label column at row_offset+t) and of bb (self mask)
  a.seg[a.nseg++] = NtxSeg{Z1, nullptr, N, row_offset, 1};
  a.seg[a.nseg++] = NtxSeg{Z2, nullptr, N, row_offset, 2};
  if (local_in_gathered) {
    // right operand of ab (rows = z1_i, LSE_a[i]; label at i == t) and of bb (rows = z2_i, LSE_b[i]; mask at i == t)
    a.seg[a.nseg++] = NtxSeg{z1, lse_a, B, 0, 1};
    a.seg[a.nseg++] = NtxSeg{z2, lse_b, B, 0, 2};
  }
  int rc = launch_bwd(a, st);
  if (rc != MAAI_OK) return rc;
  if (dz1) {
    a.P = z1; a.lse_t = lse_a; a.out = dz1; a.nseg = 0;
    a.seg[a.nseg++] = NtxSeg{Z2, nullptr, N, row_offset, 1};
    a.seg[a.nseg++] = NtxSeg{Z1, nullptr, N, row_offset, 2};
    if (local_in_gathered) {
      a.seg[a.nseg++] = NtxSeg{z2, lse_b, B, 0, 1};
      a.seg[a.nseg++] = NtxSeg{z1, lse_a, B, 0, 2};
    }
    rc = launch_bwd(a, st);
  }
  return rc;
}
