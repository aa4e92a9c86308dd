/*
 * maai_hip.h — C ABI of the MI355X-native (gfx950 / CDNA4) SimCLR hot path.
 *
 * The reference (dariodematties/Multimodal-Active-AI) is pure Python on
 * torch; it has no FFI of its own.  Its hot path reaches the device through
 * torch operators, so each entry point below names the reference call site
 * (file:line under /root/reference) whose device work it replaces.  The
 * Python mirror of the reference interface (multimodal-active-ai_amd/SimCLR/…)
 * binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host;
 *   - activations are NHWC; dtype is MAAI_BF16 (bf16 storage, fp32 MFMA
 *     accumulate) or MAAI_F32 (fp32 storage, exact fp32 MFMA — parity mode);
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises,
 *     allocates or frees (graph-capturable); workspaces come from the caller;
 *   - return 0 on success, non-zero otherwise (maai_last_error() has the text);
 *     no C++ exception crosses this boundary;
 *   - thread-safe with respect to distinct streams.
 */
#ifndef MAAI_HIP_H
#define MAAI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MAAI_BF16 0
#define MAAI_F32 1

#define MAAI_ABI_VERSION 6

int maai_abi_version(void);
const char* maai_last_error(void);
/* number of visible HIP devices, or -1 (used by the host side to fail loudly) */
int maai_device_count(void);
/* Kernel-name notes (measurement only): after maai_kernel_names(1), maai_last_kernel_name() returns the name of the device
 * kernel the calling thread's latest compute call launched last, as rocprofv3's kernel trace prints it ("" when the call
 * launched a kernel that carries no note) — bench.py's roofline.kernels rows are keyed by it.  Returns the previous setting. */
int maai_kernel_names(int on);
const char* maai_last_kernel_name(void);

/* ------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA.
 * Replaces nn.Conv2d inside resnet.py:20-28 (conv3x3/conv1x1), resnet.py:169
 * (7x7 stem), nn.Linear inside multilayerPerceptron.py:12-16 (1x1 conv over a
 * 1x1 "image"), and their autograd data gradients (same contraction with
 * transformed weights).
 *   x : [N, IH, IW, Cin]            NHWC
 *   w : [Cout, KH, KW, Cin]         K-contiguous per output channel
 *   y : [N, OH, OW, Cout]           written at (oh*out_stride+out_off_h, ow*out_stride+out_off_w)
 *                                   for oh < OHg, ow < OWg; y += result when accumulate != 0
 *   stats_partial (nullable): [maai_conv2d_stats_rows()][2][Cout] fp32 —
 *       per M-tile (64 or 128 pixels, chosen by the library): sum and sum of squares of the fp32 results per
 *       output channel (BatchNorm batch statistics, resnet.py:54,106-110,171).
 *   relu_mask (nullable): tensor laid out like y; the stored result is zeroed where relu_mask <= 0.
 *       Used by the data gradients: y is then d(loss)/d(t) for a post-ReLU tensor t = relu_mask, and the
 *       ReLU backward (resnet.py:55,111,133 under autograd) is folded into this epilogue.
 * Cin % 32 == 0 (bf16) / % 16 (f32); Cout % 64 == 0.
 * ------------------------------------------------------------------------ */
typedef struct {
  int N, IH, IW, Cin;
  int Cout, KH, KW;
  int stride, pad_h, pad_w;
  int OHg, OWg;            /* enumerated output grid */
  int OH, OW;              /* output tensor spatial size */
  int out_stride, out_off_h, out_off_w;
  int accumulate;
} maai_conv_desc;

int maai_conv2d_igemm(const maai_conv_desc* d, const void* x, const void* w, void* y, float* stats_partial,
                      const void* relu_mask, int dtype, void* stream);
long long maai_conv2d_stats_rows(const maai_conv_desc* d, int dtype);
/* Kernel family of a plain forward launch of this geometry: 0 ring / halo, 1 streaming, 2 ping-pong (8 waves, 256 x 256
 * tiles), 3 persistent 64-channel 3x3.  The families sum the statistics slab in different orders (outputs are bit-identical); a caller that needs
 * bit-reproducible statistics across input forms keeps a layer on one family. */
int maai_conv2d_kernel_family(const maai_conv_desc* d, int dtype);
/* 1 if a MAAI_EPI_BN_ACT launch (inference with frozen statistics: out = act(conv*scale + shift (+ residual)), SimCLR.py's
 * f in eval mode / the frozen backbone of SURVEY 8-f1, f4) of this geometry runs on the kernel its plain launch would use
 * (streaming, ping-pong, halo, or the ring kernel's own 128-row tile); ``lazy`` != 0: with a normalise-on-load input (xs / xt:
 * the streaming kernel only).  0: it would fall back to 128-row row-staged tiles — launch + BatchNorm pass is then faster. */
int maai_conv2d_bn_act_fast(const maai_conv_desc* d, int dtype, int lazy);

/* Fused epilogues for the HBM-bound pointwise layers: the GEMM is cheap next to its output traffic, so it is
 * run twice instead of materialising the raw conv output (resnet.py:109-110,130-133: conv3 -> bn3 -> += identity
 * -> relu):  forward = STATS_ONLY pass (no store) + BN_ACT pass (out = act(y*scale + shift (+ residual)));
 * backward = BWD_REDUCE pass (partial sums of dz and dz*(y-mean), same slab format as stats_partial) +
 * BWD_APPLY pass (dy = k1*dz - k2 - k3*y).  y is the bf16/f32-rounded accumulator, exactly the value the
 * unfused path would have stored.
 *
 * BN_ACT is not limited to pointwise layers: with frozen statistics (eval-mode BatchNorm: feature extraction for the
 * linear probe, Representation_Evaluation.py:598-712, and the frozen-backbone consumers) every convolution of the network
 * can normalise, add the shortcut and activate in its own epilogue — one launch per conv-bn-relu unit, the raw output
 * never reaches HBM.
 *
 * DGRAD_REDUCE serves the data-gradient convolutions (any kernel size, scatter and accumulate allowed): the value
 * g this launch stores — conv (+ previous content when d->accumulate) times the ReLU mask — is the gradient dz
 * entering the BatchNorm of the layer below (resnet.py:101-104,121-128: conv -> bn -> relu), so the BN-backward
 * partial sums of g and g*(t - p0) (t = that layer's raw conv output, p0 = its batch mean) are reduced here,
 * from the rounded value being stored, instead of by a separate maai_bn_act_bwd_reduce pass over dz and t.  The
 * mask is relu_mask > 0 when relu_mask is given, else (t*p1 + p2 > 0) when p1 and p2 are given (the forward
 * scale/shift of that BatchNorm: its ReLU output is positive exactly there, so the mask tensor need not be read),
 * else none.  stats_partial receives maai_conv2d_stats_rows(d) rows of [2][Cout]. */
#define MAAI_EPI_STORE 0
#define MAAI_EPI_STATS_ONLY 1
#define MAAI_EPI_BN_ACT 2
#define MAAI_EPI_BWD_REDUCE 3
#define MAAI_EPI_BWD_APPLY 4
#define MAAI_EPI_DGRAD_REDUCE 5
typedef struct {
  int mode;
  int relu;          /* BN_ACT: apply max(.,0) */
  const float* p0;   /* BN_ACT: scale (NULL = 1); BWD_REDUCE, DGRAD_REDUCE: mean (NULL = 0); BWD_APPLY: k1 */
  const float* p1;   /* BN_ACT: shift (NULL = 0); BWD_APPLY: k2; DGRAD_REDUCE: scale of the mask (nullable) */
  const float* p2;   /* BWD_APPLY: k3; DGRAD_REDUCE: shift of the mask (nullable) */
  const void* t;     /* BN_ACT: residual laid out like y (nullable); BWD_*: dz laid out like y;
                        DGRAD_REDUCE: the lower layer's raw conv output, laid out like y */
  int mask_bits;     /* DGRAD_REDUCE, bf16: relu_mask is the 1-bit mask of maai_bn_act_fwd_mask, not a tensor */
  int sum_increment; /* DGRAD_REDUCE with d->accumulate: the partial sums are those of (stored value - previous content), i.e.
                        of what this launch changes.  A strided second pass over a tensor whose first, dense pass reduced its
                        own values (stride-2 downsample gradient added to the conv1 gradient) then completes the sums of the
                        final tensor — pixels it does not touch contribute nothing — without a separate reduction pass */
  /* DGRAD_REDUCE on a pointwise bf16 layer, optional: the A operand of the GEMM is not x itself but
   * k1[c]*x - k2[c] - k3[c]*a2 per input channel c, i.e. the BatchNorm-backward apply of the layer whose gradient
   * this convolution propagates (x = dz, a2 = that layer's raw conv output), formed while staging instead of by a
   * maai_bn_act_bwd_apply pass that writes it and this launch reading it back.  a_out (nullable) receives the
   * transformed operand [M][Cin] for the weight gradient.  NULL a2 = plain operand. */
  const void* a2;
  const float* ak1;
  const float* ak2;
  const float* ak3;
  void* a_out;
  /* Normalise-on-load (mode STORE, dense, non-accumulating forward launches; any kernel size / stride):
   * the A operand of the GEMM is not x itself but  act(x*xs[ci] + xt[ci])  per input channel ci, i.e. x is the RAW
   * convolution output of the unit below and (xs, xt) its BatchNorm scale / shift (resnet.py:101-109: conv2 consumes
   * relu(bn1(conv1(x))), conv3 consumes relu(bn2(..))).  The transform is applied to the staged operand in LDS, with
   * the arithmetic of maai_bn_act_fwd (fp32 multiply, add, max, one rounding to the storage type) and with zero
   * padding applied AFTER it, so the result is bit-identical to maai_bn_act_fwd followed by the plain launch — and
   * the normalised activation never exists in HBM.  x_relu != 0 applies max(.,0).  NULL xs = plain operand.
   * Pointwise stride-1 layers may join two raw tensors (resnet.py:126-133, `relu(bn3(y3) + identity)` consumed by
   * the next block's conv1): A = act((x*xs + xt) + r(xb*xs2 + xt2)), xb laid out like x, r = rounding to the storage
   * type (xs2 == NULL: xb is added as is — an identity shortcut), bit-identical to maai_bn_act_fwd2 / maai_bn_act_fwd
   * with a residual.  x_out (nullable, laid out like x) receives the joined activation and x_bits (nullable, bf16
   * only) its 1-bit ReLU mask in the format of maai_bn_act_fwd_mask, each written exactly once. */
  const float* xs;
  const float* xt;
  int x_relu;
  const void* xb;
  const float* xs2;
  const float* xt2;
  void* x_out;
  unsigned char* x_bits;
  /* Chained launch (bf16, pointwise, with the two-tensor join above; resnet.py:118-133 + :101 of the next block): x is
   * not read at all.  It is RECOMPUTED as the pointwise convolution  conv(a, pre_w)  of pre_x [M][pre_cin] (a = pre_x, or
   * act(pre_x*pre_xs + pre_xt) when pre_xs is given: normalise-on-load of conv3's own input) with pre_w [Cin][pre_cin],
   * rounded to the storage type exactly as the launch that would have stored it; (xs, xt) are then the BatchNorm
   * coefficients of that recomputed tensor — whose batch statistics the caller has taken with a MAAI_EPI_STATS_ONLY
   * launch of the same convolution — and the join, x_out, x_bits and this convolution proceed as above, bit-identical to
   * the unchained sequence.  pre_y_out (nullable, [M][Cin]) receives the recomputed tensor when the backward pass needs
   * it.  Built for pre_cin = 64, Cin = 256; x_out is required.  NULL pre_x = not chained. */
  const void* pre_x;
  const void* pre_w;
  const float* pre_xs;
  const float* pre_xt;
  int pre_relu;
  int pre_cin;
  void* pre_y_out;
  /* two-source input (MAAI_EPI_DGRAD_REDUCE on pointwise stride-1 bf16 layers: the data gradient of a unit whose BatchNorm
   * backward is folded through its convolution, see maai_fold_dgrad_w): input channels [0, cin1) are read from x (rows of
   * cin1 elements), [cin1, Cin) from x2 (rows of Cin - cin1); both multiples of 64.  bias (nullable, [Cout]) is added to the
   * accumulators before they are rounded and stored (also without x2). */
  const void* x2;
  int cin1;
  const float* bias;
  /* diag (nullable, [Cout]; with p1/p2 = the lower unit's scale/shift and t = its raw output): out[p][c] += diag[c] * a[p][c]
   * with a = relu(r(t*scale + shift)), the lower unit's activation recomputed in the epilogue — T's diagonal of the folded
   * data gradient in fp32 (maai_fold_dgrad_w, dg). */
  const float* diag;
} maai_conv_epilogue;
int maai_conv2d_igemm_fused(const maai_conv_desc* d, const void* x, const void* w, void* y, float* stats_partial,
                            const void* relu_mask, const maai_conv_epilogue* epi, int dtype, void* stream);
long long maai_conv2d_stats_rows_fused(const maai_conv_desc* d, const maai_conv_epilogue* epi, int dtype);

/* Backward of a 64 -> 256 bottleneck's last unit in one launch (bf16; resnet.py:118-119, autograd of conv3 + bn3):
 *   dz = k1*g - k2 - k3*y3 (BatchNorm-backward apply; g, y3: [M][256]),
 *   dx[M][64] = (dz . W3 (+ dx's previous content when `accumulate`)) masked by y2*s2 + t2 > 0 (the ReLU of conv3's input
 *              a2 = relu(bn2(y2)), y2: [M][64]),
 *   slab[rows][2][64]: per-workgroup sums of dx and dx*(y2 - mean2) (bn2's backward sums; rows = maai_conv_bwd3_rows(M)),
 *   dw[256][64] (fp32, zeroed by the caller) += dz^T . a2.
 * wd: conv3's weights as [64][1][1][256] (data-gradient form).  dz is never written to memory.  dx and the sums follow
 * maai_conv2d_igemm_fused's DGRAD_REDUCE epilogue with a2 (same roundings). */
int maai_conv_bwd3_rows(long long M);
int maai_conv_bwd3(const void* g, const void* y3, const void* y2, const void* wd, const float* k1, const float* k2, const float* k3,
                   const float* mean2, const float* s2, const float* t2, void* dx, float* slab, float* dw, long long M, int accumulate,
                   void* stream);

/* Weight gradient of the same convolution (autograd of nn.Conv2d / nn.Linear):
 *   dw[co][kh][kw][ci] += sum_m dy[m][co] * x[n, oh*s-ph+kh, ow*s-pw+kw, ci]
 * dw is fp32 [Cout][KH][KW][Cin] and must be zeroed by the caller (split-K
 * partial products are added with fp32 atomics). dy is dense [N,OH,OW,Cout]. */
int maai_conv2d_wgrad(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype, void* stream);
/* Same, with the split-K workgroup budget as a tuning knob (0 = default): more workgroups hide latency,
 * fewer add less fp32-atomic traffic; the host side measures once per shape and caches the choice. */
int maai_conv2d_wgrad_tuned(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype,
                            int target_blocks, void* stream);
/* Same, with the x operand normalised on load: x is the RAW convolution output of the unit below and the gradient is
 * taken with respect to the weights that multiplied act(x*xs[ci] + xt[ci]) (maai_conv_epilogue.xs/xt/x_relu: the
 * forward pass never stored that activation).  Zero padding is applied after the transform.  Bit-identical to
 * maai_bn_act_fwd followed by maai_conv2d_wgrad_tuned up to the order of the fp32 atomic adds.  xs == NULL: plain. */
int maai_conv2d_wgrad_xf(const maai_conv_desc* d, const void* x, const void* dy, float* dw, int dtype, int target_blocks,
                         const float* xs, const float* xt, int x_relu, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm (training) — nn.BatchNorm2d / nn.SyncBatchNorm as norm_layer
 * (resnet.py:54,57,106-110,171,212; Contrastive_Learning.py:240-252).
 * ------------------------------------------------------------------------ */
/* sums[2][C] (double) = column sums of partial[rows][2][C] (fp32) */
int maai_reduce_partials(const float* partial, long long rows, int C2, double* sums, void* stream);
/* From (possibly all-reduced) sums over `count` samples: mean, biased var,
 * invstd = rsqrt(var+eps); scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var updated with momentum (unbiased var), all fp32 [C].
 * running_* may be NULL (no update). */
int maai_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                     float* shift, int C, void* stream);
/* SyncBatchNorm exchange in fp32 (nn.SyncBatchNorm as selected at Contrastive_Learning.py:240-252; torch gathers
 * mean | invstd | count per layer): maai_bn_pack_stats turns a rank's fp64 sums over `count` samples into
 * packed[2C+1] = mean[C] | M2[C] = sum (x - mean)^2 | count (bit pattern of an int32); after an all-gather,
 * maai_bn_finalize_gathered merges the `world` rows (row r at gathered + r*row_stride, row_stride >= 2C+1 floats)
 * with Chan's parallel-variance formula in fp64 and produces what maai_bn_finalize produces; count_out (nullable, device)
 * receives the merged sample count N = sum of the rows' counts — the ranks' batches may differ — for the backward's 1/N
 * (maai_bn_bwd_coeffs*, count_dev), as nn.SyncBatchNorm uses the summed gathered counts in both passes. */
int maai_bn_pack_stats(const double* sums, double count, float* packed, int C, void* stream);
int maai_bn_finalize_gathered(const float* gathered, int world, long long row_stride, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                              float* scale, float* shift, double* count_out, int C, void* stream);
/* Deferred running-statistic updates: slots[i] = {running [n], stat_a [n], stat_b [n] or null, n, momentum}: running <-
 * (1 - momentum)*running + momentum*stat_a, then the same with stat_b — maai_bn_finalize's arithmetic, applied for every layer
 * in ONE launch after two forwards that ran concurrently (each wrote its (float)mean / (float)unbiased variance to scratch:
 * maai_bn_finalize with momentum 1 on zeroed buffers).  slots: device memory. */
typedef struct {
  float* running;
  const float* stat_a;
  const float* stat_b;
  long long n;
  float momentum;
  int reserved;
} maai_bn_update_slot;
int maai_bn_running_update_multi(const maai_bn_update_slot* slots, int nslots, void* stream);
/* eval mode: scale/shift from running statistics */
int maai_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* scale, float* shift, int C, void* stream);
/* out = act(y*scale[c] + shift[c] (+ residual)); scale/shift nullable (1 / 0);
 * relu != 0 applies max(.,0).  resnet.py:54-55,72-75,106-133; bias+ReLU of
 * multilayerPerceptron.py:13-14 with scale == NULL. */
int maai_bn_act_fwd(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                    long long M, int C, int relu, int dtype, void* stream);
/* same, and (bf16 only, mask_bits nullable) the 1-bit-per-element ReLU mask of the stored output: byte i covers
 * elements 8i..8i+7 of the NHWC tensor, bit e set where out > 0.  The backward pass of a residual block reads
 * this (M*C/8 bytes) instead of the block output itself to mask the gradient (resnet.py:133 relu backward);
 * consumed by maai_conv2d_igemm_fused with epi->mask_bits = 1. */
int maai_bn_act_fwd_mask(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                         unsigned char* mask_bits, long long M, int C, int relu, int dtype, void* stream);
/* Two normalised branches in one pass (first block of a stage, resnet.py:126-133 with a downsample):
 * out = act((y*scale + shift) + r(y2*scale2 + shift2)), r = rounding to the storage type, so the result is
 * bit-identical to maai_bn_act_fwd(y2 -> idn) followed by maai_bn_act_fwd(y, residual = idn) without ever
 * storing idn.  mask_bits as above (nullable).  The backward counterpart reads the shared gradient once:
 * dy = k1*dz - k2 - k3*y and dy2 = k1b*dz - k2b - k3b*y2. */
int maai_bn_act_fwd2(const void* y, const float* scale, const float* shift, const void* y2, const float* scale2,
                     const float* shift2, void* out, unsigned char* mask_bits, long long M, int C, int relu, int dtype,
                     void* stream);
int maai_bn_act_bwd_apply2(const void* dz, const void* y, const float* k1, const float* k2, const float* k3,
                           const void* y2, const float* k1b, const float* k2b, const float* k3b, void* dy, void* dy2,
                           long long M, int C, int dtype, void* stream);
/* backward pass 1: dz = dout * (out > 0 if relu); partial[rows][2][C] = per-block
 * column sums of dz and dz * (y - mean[c]).  rows = maai_bn_bwd_rows(M, C, dtype).
 * y/mean nullable (second sum left 0), out nullable when relu == 0.  Also the
 * bias gradient of nn.Linear (column sum of dout). */
long long maai_bn_bwd_rows(long long M, int C, int dtype);
int maai_bn_act_bwd_reduce(const void* dout, const void* out, const void* y, const float* mean, float* partial,
                           long long M, int C, int relu, int dtype, void* stream);
/* From sums = [S1 = sum dz, S2 = sum dz*(y-mean)] (double, possibly all-reduced) over `count`:
 *   dbeta = S1, dgamma = invstd*S2,
 *   dy = k1*dz - k2 - k3*y,  k1 = gamma*invstd, k3 = k1*invstd^2*S2/count, k2 = k1*S1/count - k3*mean
 * count_dev (nullable): a device double that overrides `count` (the merged count of maai_bn_finalize_gathered). */
int maai_bn_bwd_coeffs(const double* sums, double count, const float* gamma, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* k1, float* k2, float* k3, int C, const double* count_dev, void* stream);
/* the same from fp32 sums (the cross-rank all-reduce of SyncBatchNorm's backward travels in fp32) */
int maai_bn_bwd_coeffs_f32(const float* sums, double count, const float* gamma, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* k1, float* k2, float* k3, int C, const double* count_dev,
                           void* stream);
/* backward pass 2: dz = dout*(out>0); dy = k1[c]*dz - k2[c] - k3[c]*y; optional
 * dz_out (gradient of the residual input) written too. k* nullable -> dy = dz. */
int maai_bn_act_bwd_apply(const void* dout, const void* out, const void* y, const float* k1, const float* k2,
                          const float* k3, void* dy, void* dz_out, long long M, int C, int relu, int dtype,
                          void* stream);

/* ------------------------------------------------------------------------
 * View packing / layout (SimCLR.py:24) and pooling (resnet.py:181 variant)
 * ------------------------------------------------------------------------ */
/* K views [B,H,W,3] u8 (HWC) -> NHWC [B,H,W,Cpad], channel k*3+c, raw 0..255,
 * zero padded to Cpad.  views: HOST array of K device pointers (K <= 8). */
int maai_pack_views_u8(const void* const* views_host, int K, int B, int H, int W, int Cpad, void* out, int dtype,
                       void* stream);
/* Stem operand for 3-channel input: out[n,h,w, kw*4+c] = x[n,h,w+kw-3,c] (kw<7,c<3; else 0)
 * from NCHW fp32 x [B,3,H,W] or from one u8 HWC view; turns the 7x7 stem into a 7x1 conv with Cin = 32. */
int maai_stem_unroll_nchw_f32(const float* x, int B, int H, int W, void* out, int dtype, void* stream);
int maai_stem_unroll_u8(const void* view, int B, int H, int W, void* out, int dtype, void* stream);
/* NCHW fp32 <-> NHWC T (channel zero-padding on the way in, dropped on the way out) */
int maai_nchw_f32_to_nhwc(const float* x, int B, int C, int H, int W, int Cpad, void* out, int dtype, void* stream);
int maai_nhwc_to_nchw_f32(const void* x, int B, int C, int H, int W, int Cpad, float* out, int dtype, void* stream);
int maai_nchw_f32_from_nhwc_grad(const void* g, int B, int C, int H, int W, int Cpad, float* out, int dtype, void* stream);
/* adaptive average pool to PH x PW windows (H % PH == 0, W % PW == 0) */
int maai_avgpool_fwd(const void* x, int B, int H, int W, int C, int PH, int PW, void* out, int dtype, void* stream);
int maai_avgpool_bwd(const void* dout, int B, int H, int W, int C, int PH, int PW, void* dx, int dtype, void* stream);
/* dtype casts: fp32 <-> T, n elements */
int maai_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream);
int maai_cast_to_f32(const void* src, float* dst, long long n, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * NT-Xent (Objective.py:17-81,123-125), fp32 throughout, MFMA fp32.
 * h1,h2 [B,d] raw embeddings of this rank; Z1,Z2 [N,d] L2-normalised
 * embeddings of all ranks (== this rank's when world == 1); row_offset =
 * rank*B (Objective.py:55).
 * ------------------------------------------------------------------------ */
/* z = h / max(||h||, 1e-12) row-wise (Objective.py:42-43); inv_norm [B] saved for backward.
 * normalize == 0 copies. */
int maai_ntxent_normalize(const float* h, float* z, float* inv_norm, int B, int d, int normalize, void* stream);
/* loss (1 float, zeroed by callee), logits_ab [B,N] (= z1 Z2^T / tau), lse [2][B]
 * (row log-sum-exp of [ab,aa] and [ba,bb] with the -1e9 self mask). d % 16 == 0, d <= 496 (the
 * backward pass keeps 16 x (d + 4) + 64 x d floats in LDS: 160 KiB). */
int maai_ntxent_fwd(const float* z1, const float* z2, const float* Z1, const float* Z2, float* loss, float* logits_ab,
                    float* lse, int B, int N, int d, float temperature, int row_offset, void* stream);
/* gradients of loss*gloss wrt z2 (always) and z1 (nullable).  local_in_gathered != 0
 * means Z1/Z2 rows [row_offset, row_offset+B) are differentiable aliases of z1/z2
 * (single-process semantics); 0 = gathered copies are constants (Objective.py:112-114). */
int maai_ntxent_bwd(const float* z1, const float* z2, const float* Z1, const float* Z2, const float* lse,
                    const float* gloss, float* dz1, float* dz2, int B, int N, int d, float temperature, int row_offset,
                    int local_in_gathered, void* stream);
/* through the normalisation: dh = (dz - z * (z . dz)) * inv_norm */
int maai_ntxent_normalize_bwd(const float* z, const float* dz, const float* inv_norm, float* dh, int B, int d,
                              int normalize, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm-backward folded through a channel-expanding pointwise convolution (csrc/fold.hip): y = x W^T followed by
 * training-mode BatchNorm — conv3 + bn3 of the reference's Bottleneck (SimCLR/ResNet/resnet.py:118-119).  With
 * G1 = g^T x (maai_conv2d_wgrad on the UN-normalised gradient g), Gram = x^T x and sx = colsum(x):
 *   maai_fold_s2:      s2[c] = sum_k W[c,k] G1[c,k] - mean[c] s1[c]                 (= sum g*(y - mean) without reading y)
 *   maai_fold_dw:      dw[c,k] = k1[c] G1[c,k] - k2[c] sx[k] - k3[c] (W Gram)[c,k]  (= (k1*g - k2 - k3*y)^T x)
 *   maai_fold_dgrad_w: wf[k][c] = bf16(k1[c] W[c,k]), tn[k][j] = bf16(-(W^T diag(k3) W)[j,k]), cn[k] = -(k2 W)[k] - comp[k], so that
 *                      dx = g wf^T + x tn^T + cn                                    (= (k1*g - k2 - k3*y) W)
 *                      comp[k] = (s1 . (wf[k] - exact) + sx . (tn[k] - exact)) / count: the pixel mean of what the bf16 rounding
 *                      of the folded weights adds to dx[:, k] (s1 = sum g, sx = colsum x over `count` pixels), taken out of the
 *                      constant.  dg (nullable, [Cin] fp32): the DIAGONAL of -(W^T diag(k3) W) — a sum of squares, the one large
 *                      entry per row, multiplying the very x[p][k] the ReLU mask and the unit below's BatchNorm-backward sums
 *                      are made of — is then returned in fp32 and tn[k][k] = 0: the data-gradient launch adds dg[k]*x[p][k] in its
 *                      epilogue (maai_conv_epilogue.diag / maai_conv_dfold).  scratch: Cin floats.
 * w: the bf16 kernel-layout weights [Cout][Cin]; g1, gram, dw fp32; s1, s2, sx fp64; k1..k3 as maai_bn_bwd_coeffs gives them.
 * ------------------------------------------------------------------------ */
int maai_fold_s2(const void* w, const float* g1, const double* s1, const float* mean, double* s2, int Cout, int Cin, void* stream);
int maai_fold_dw(const void* w, const float* g1, const float* gram, const double* sx, const float* k1, const float* k2,
                 const float* k3, float* dw, int Cout, int Cin, void* stream);
int maai_fold_dgrad_w(const void* w, const float* k1, const float* k2, const float* k3, const double* s1, const double* sx,
                      double count, void* wf, int wf_pitch, void* tn, int tn_pitch, float* cn, float* dg, float* scratch, int Cout, int Cin,
                      void* stream);
/* wf_pitch / tn_pitch: row pitches (elements) of wf [Cin rows] and tn [Cin rows]: Cout and Cin for two separate matrices, or
 * both Cout + Cin with tn = wf + Cout for the concatenated form Wcat [Cin][Cout + Cin] that maai_conv_dfold multiplies by.
 *
 * maai_gram (csrc/gram.hip): gram [C][C] fp32 += x^T x, sx [C] fp64 += colsum(x), npos [C] fp64 (nullable) += rows with x > 0, over
 * the M rows of x [M][C] bf16 (C = 64, 128, 256, 512); xs / xt (nullable, [C]) + x_relu: x is act(raw*xs + xt) formed on load
 * (maai_bn_act_fwd arithmetic).
 * maai_conv_dfold (csrc/conv_dfold.hip): the folded unit's data gradient for the 64 -> 256 units of layer 1 in one launch —
 *   dx[M][64] = ([g | a2] Wcat^T + cn + dg*a2) * [a2 > 0],  a2 = relu(r(y2*s2 + t2)) formed on load from the raw y2 [M][64], g [M][256],
 *   (+)= when accumulate; slab [maai_conv_dfold_rows(M)][2][64]: partial sums of dx and dx*(y2 - mean2) (the unit below's
 *   BatchNorm backward). */
int maai_gram(const void* x, long long M, int C, const float* xs, const float* xt, int x_relu, float* gram, double* sx, double* npos,
              void* stream);
/* Deterministic form: pgram [maai_gram_partial_rows(M, C)][C][C] and psx [rows][C] (fp32) are WRITTEN, one row per workgroup
 * column, no atomics; summed by maai_reduce_partials they are bit-reproducible and may feed forward statistics:
 * maai_fold_stats: sums[c] = W[c] . sx, sums[Cout + c] = W[c] Gram W[c]^T (fp64; gram [Cin][Cin], sx [Cin] fp64) — the BatchNorm
 * statistics of y = x W^T without computing y, in the layout maai_bn_finalize takes (the chained block boundaries of layer 1). */
int maai_gram_partial_rows(long long M, int C);
int maai_gram_partials(const void* x, long long M, int C, const float* xs, const float* xt, int x_relu, float* pgram, float* psx,
                       void* stream);
int maai_fold_stats(const void* w, const double* gram, const double* sx, double* sums, int Cout, int Cin, void* stream);
long long maai_conv_dfold_rows(long long M);
int maai_conv_dfold(const void* g, const void* y2, const void* w, const float* cn, const float* dg, const float* mean2, const float* s2,
                    const float* t2, void* dx, float* slab, long long M, int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * Comm helper: one-shot direct all-gather over the xGMI mesh (symmetric buffers, peer-to-peer stores) — the
 * transport-level replacement of Objective._cross_replica_concat's dist.all_gather + torch.cat (SimCLR/Objective.py:
 * 102-114) for the [B,128] embeddings; the RCCL path (torch.distributed) remains the default.  Setup: every rank
 * creates its communicator (max_bytes = largest message, % 4 == 0), sends the 64-byte handle of its buffer to every peer
 * by any means (the callers use their process group) and attaches the handles it receives.  maai_comm_allgather then
 * writes this rank's message into its slot of every peer's buffer (8-byte {epoch, payload} granules, one system-scope
 * store each: the data is the flag) and gathers dst[world][bytes] from its own buffer, all in one kernel on `stream`.
 * A rank may run at most one gather ahead of any other rank (two epoch parities).  The buffers are fine-grained device
 * memory (maai_comm_create fails where that is unavailable: no coarse-grained fallback).  A sweep waits for a granule at most
 * MAAI_P2P_TIMEOUT_MS of wall-clock time (default 120 000); a gather that gives up delivers quiet NaNs for the granules it
 * did not get (never stale data) and sets a host-visible status word to its epoch: maai_comm_poll returns that word without
 * synchronising, maai_comm_status after hipDeviceSynchronize, and maai_comm_allgather refuses (MAAI_ERR_LAUNCH) once it is
 * set — the ranks' epochs have diverged and the communicator is dead.
 * ------------------------------------------------------------------------ */
typedef struct maai_comm maai_comm;
int maai_comm_create(int rank, int world, long long max_bytes, maai_comm** out);
int maai_comm_handle(maai_comm* c, void* handle64);
int maai_comm_attach(maai_comm* c, int peer, const void* handle64);
int maai_comm_allgather(maai_comm* c, const void* src, long long bytes, void* dst, void* stream);
int maai_comm_status(maai_comm* c, unsigned* status);
int maai_comm_poll(maai_comm* c, unsigned* status);
int maai_comm_destroy(maai_comm* c);

/* ------------------------------------------------------------------------
 * Optimiser (Model_Util.py:68-88: torch.optim.Adam / SGD)
 * ------------------------------------------------------------------------ */
int maai_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                   double eps, int step, float grad_scale, void* stream);
int maai_sgd_step(float* p, const float* g, float* mom, long long n, float lr, float momentum, float weight_decay,
                  int first_step, void* stream);
/* Kernel-layout copies of the fp32 master convolution weights [Cout][Cin][KH][KW], any number of layers and forms in
 * one launch.  mode 0 (forward operand): out[co][a][b][ci] = w[co][ci][khs[a]][kws[b]] for ci < Cin, 0 up to cin_pad;
 * mode 1 (data-gradient operand): out[ci][a][b][co] = w[co][ci][khs[a]][kws[b]].  forms[], block_form[], block_first[]
 * are device arrays: block b converts 1024 consecutive output elements of forms[block_form[b]] from block_first[b]. */
typedef struct {
  const float* w;
  void* out;
  int Cout, Cin, KH, KW;
  int mode;
  int nkh, nkw, cin_pad;
  int khs[8], kws[8];
  int dtype;
  int reserved;
} maai_weight_form;
int maai_weight_forms(const maai_weight_form* forms, const int* block_form, const long long* block_first, int nblocks,
                      void* stream);
/* the same Adam update for every tensor of a parameter group in one launch (they share lr / betas / eps / step):
 * slots[] (device memory) lists the tensors; block b of the launch updates 2048 consecutive elements of tensor
 * block_slot[b] starting at element block_first[b] (both device arrays of nblocks entries, built by the host from
 * the tensor sizes).  Identical arithmetic to maai_adam_step. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
} maai_adam_slot;
int maai_adam_step_multi(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                         double lr, double beta1, double beta2, double eps, int step, float grad_scale, void* stream);
/* torch.optim.SGD (Model_Util.py:70-73) for every tensor of a parameter group in one launch: slots as above with m = the
 * momentum buffer (v unused); identical arithmetic to maai_sgd_step. */
int maai_sgd_step_multi(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                        float lr, float momentum, float weight_decay, int first_step, void* stream);

/* LARC (Model_Util.py:80-83: `--optimizer lars` = apex.parallel.LARC around Adam; Apex's published algorithm,
 * trust_coefficient 0.02, clip mode — parity UNPINNED, Apex cannot be installed here).  slots / block maps as in
 * maai_adam_step_multi (m and v unused).  maai_multi_sqnorm: norms[2*s] = ||p_s||^2, norms[2*s+1] = ||g_s||^2 (fp64,
 * zeroed by the callee) for every tensor in ONE launch.  maai_larc_scale: for every tensor with non-zero norms,
 * g <- (g + wd*p) * rate, rate = trust*||p|| / (||g|| + wd*||p|| + eps), clipped to min(rate/lr, 1) when clip. */
int maai_multi_sqnorm(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                      int nslots, double* norms, void* stream);
int maai_larc_scale(const maai_adam_slot* slots, const int* block_slot, const long long* block_first, int nblocks,
                    const double* norms, float trust_coefficient, float lr, float weight_decay, float eps, int clip,
                    void* stream);

/* ------------------------------------------------------------------------
 * Linear probe on frozen features (Representation_Evaluation.py:621-666): the logits are a 1x1 convolution
 * (maai_conv2d_igemm, fp32); this is nn.CrossEntropyLoss on them — mean over the batch of
 * logsumexp(logits[b,:C]) - logits[b,label[b]], class-index targets (int64) — and its gradient
 * dlogits[b,c] = (softmax - onehot) * gloss / B.  logits / dlogits are [B][ld] with ld >= C (columns C..ld-1 are
 * padding of the GEMM's 64-column granularity: ignored, gradient 0).  loss (1 float) is zeroed by the callee;
 * lse [B] is kept for the backward pass.
 * ------------------------------------------------------------------------ */
int maai_softmax_ce_fwd(const float* logits, const long long* labels, float* loss, float* lse, int B, int C, int ld,
                        void* stream);
int maai_softmax_ce_bwd(const float* logits, const long long* labels, const float* lse, const float* gloss, float* dlogits,
                        int B, int C, int ld, void* stream);

/* ------------------------------------------------------------------------
 * Two-view augmentation replacing NVIDIA_DALI_Pipelines.py:444-480 for the
 * north-star path: crop window -> nearest resize -> flip -> colour twist (brightness, contrast, then hue rotation
 * and saturation as ONE 3x3 colour matrix in DALI ColorTwist's YIQ form).
 * images [B,H,W,3] u8; params [B][16] f32 = x0,y0,cw,ch,flip,brightness,contrast, M[3][3] (row major);
 * out [B,OH,OW,3] u8:  v = ((src - 128)*contrast + 128)*brightness per channel, out = round(clamp(M v, 0, 255)).
 * ------------------------------------------------------------------------ */
int maai_augment_view_u8(const void* images, const float* params, int B, int H, int W, int OH, int OW, void* out,
                         void* stream);
/* Philox-free counter hash: fills params for `B` samples from (seed, view) per
 * Contrastive_Learning.py:601-635's ranges: brightness = (1 - b/2) + b*u, contrast likewise, hue = u*`hue` degrees,
 * saturation = (1 - s) + s*u; M = YIQ2RGB * R(hue) * diag(1, sat, sat) * RGB2YIQ. */
int maai_augment_params(float* params, int B, int H, int W, unsigned long long seed, int view, float min_area,
                        float brightness, float contrast, float saturation, float hue, void* stream);

/* ------------------------------------------------------------------------
 * Foveated retinal processor: the whole DALI graph of NVIDIA_DALI_Pipelines.py:444-480
 * (RandomResizedCrop 640 -> Rotate -> GridMask -> noise -> Flip -> ColorTwist -> crops 400/240/100/30 ->
 * resize 30x30) as ONE kernel.  images [B,H,W,3] u8 (padded batch; per-sample extents in params),
 * params [B][32] f32 (layout in csrc/foveate.hip), out [4][B][OS][OS][3] u8 (view-major).
 * ------------------------------------------------------------------------ */
int maai_foveate_views_u8(const void* images, const float* params, int B, int H, int W, int OS, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAAI_HIP_H */
