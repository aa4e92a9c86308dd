"""Execution engine of the SimCLR hot path on the HIP kernels.

The reference runs ``g(f(x))`` through torch autograd over ~160 small modules
(SimCLR.py:23-31, resnet.py:226-240, multilayerPerceptron.py:18-22).  Here the
whole backbone (+ head) is ONE autograd node: the forward is an explicit
sequence of conv(+BN statistics) / BN-apply launches on NHWC tensors and the
backward is the hand-written reverse sequence (BN reduce -> BN apply -> weight
gradient -> data gradient, residual gradients accumulated in the conv
epilogue), so there is no per-op autograd bookkeeping, no NCHW<->NHWC traffic
between layers and every saved tensor is explicit.

Parameters stay ordinary fp32 ``nn.Parameter`` s in the reference layout
(state_dict compatible); their bf16 / KHWC device copies are
cached per optimiser step.
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import kernels as K
from ._lib import MaaiError

_PRECISION = {"dtype": torch.bfloat16}
_WEIGHT_EPOCH = [0]
# Weight gradients are independent of the data-gradient chain: they run on a side HIP stream so that the
# MFMA/atomic-bound wgrad kernels overlap the HBM-bound BatchNorm passes of the following layers.
_SIDE = {"enabled": os.environ.get("MAAI_WGRAD_SIDE_STREAM", "0") == "1", "stream": None}  # measured +0.7 %: off by default


def _side_stream():
    if _SIDE["stream"] is None:
        _SIDE["stream"] = torch.cuda.Stream()
    return _SIDE["stream"]


def set_precision(name):
    """'bf16' (production: bf16 storage, fp32 MFMA accumulate) or 'fp32'
    (fp32 storage + exact-fp32 MFMA: the parity mode)."""
    if name not in ("bf16", "fp32"):
        raise ValueError("precision must be 'bf16' or 'fp32'")
    _PRECISION["dtype"] = torch.bfloat16 if name == "bf16" else torch.float32


def get_precision():
    return "bf16" if _PRECISION["dtype"] == torch.bfloat16 else "fp32"


def compute_dtype():
    return _PRECISION["dtype"]


def bump_weight_epoch():
    """Called by the HIP optimisers after they rewrite parameters through raw pointers."""
    _WEIGHT_EPOCH[0] += 1


# ----------------------------------------------------------------------------
# weight forms: reference layout (fp32) <-> kernel layout (storage dtype)
# ----------------------------------------------------------------------------
_CACHE = {}


def _cached(param, tag, dtype, build):
    key = (id(param), tag, dtype)
    ver = (param._version, _WEIGHT_EPOCH[0], param.data_ptr())
    hit = _CACHE.get(key)
    # (the entry must belong to THIS parameter object: a destroyed model's id and allocator address can both be
    #  recycled by the next one, with _version 0 again)
    if hit is not None and hit[0] == ver and hit[2]() is param:
        return hit[1]
    with torch.no_grad():
        t = build(param.detach())
    if len(_CACHE) > 64:
        for k in [k for k, h in _CACHE.items() if h[2]() is None]:
            del _CACHE[k]
    _CACHE[key] = (ver, t, weakref.ref(param))
    return t


def clear_weight_cache():
    _CACHE.clear()
    _FORMS["reg"] = None
    _FORMS["index"].clear()


def _cast(t, dtype):
    t = t.contiguous()
    return K.cast_from_f32(t, dtype) if dtype != torch.float32 else t


# Kernel-layout copies of the convolution weights go through one registry: a copy requested for the first time is
# converted on its own; once any copy is found stale (the optimiser stepped), ALL registered copies are refreshed
# by a single launch (kernels.WeightForms) — ~400 tiny torch kernels per step otherwise.
_FORMS = {"reg": None, "index": {}}


def _form(param, dtype, tag, mode, khs, kws, cin_pad=None):
    if _FORMS["reg"] is None:
        _FORMS["reg"] = K.WeightForms()
    reg = _FORMS["reg"]
    key = (id(param), tag, dtype)
    ver = (param._version, _WEIGHT_EPOCH[0], param.data_ptr())
    hit = _FORMS["index"].get(key)
    if hit is not None and hit[3]() is not param:
        hit = None   # the id was recycled by a new parameter object
    if hit is None:
        _purge_forms()
        reg = _FORMS["reg"]
        with torch.no_grad():
            idx, out = reg.add(param.detach(), dtype, mode, khs, kws, cin_pad)
            reg.run([idx])
        _FORMS["index"][key] = [ver, out, idx, weakref.ref(param)]
        return out
    if hit[0] != ver:
        with torch.no_grad():
            reg.run()   # everything registered, one launch
        for h in _FORMS["index"].values():
            q = h[3]()
            if q is not None:
                h[0] = (q._version, _WEIGHT_EPOCH[0], q.data_ptr())
    return hit[1]


def _purge_forms():
    """Drop the copies of parameters that no longer exist (models come and go in tests and sweeps)."""
    idx = _FORMS["index"]
    dead = [k for k, h in idx.items() if h[3]() is None]
    if not dead:
        return
    for k in dead:
        del idx[k]
    old = _FORMS["reg"]
    reg = K.WeightForms()
    for h in idx.values():
        reg.entries.append(old.entries[h[2]])
        h[2] = len(reg.entries) - 1
    _FORMS["reg"] = reg


def w_fwd(param, dtype, cin_pad=None):
    """[Cout,Cin,KH,KW] -> [Cout,KH,KW,Cin(_pad)]"""
    if param.is_cuda and param.dim() == 4 and param.is_contiguous() and param.shape[2] <= 8 and param.shape[3] <= 8:
        return _form(param, dtype, ("fwd", cin_pad), 0, list(range(param.shape[2])), list(range(param.shape[3])), cin_pad)

    def build(w):
        w = w.permute(0, 2, 3, 1)
        if cin_pad is not None and cin_pad != w.shape[3]:
            w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[3]))
        return _cast(w, dtype)
    return _cached(param, ("fwd", cin_pad), dtype, build)


def w_stem_unrolled(param, dtype):
    """[64,3,7,7] -> [64,7,1,32] with channel kw*4+c (elementwise.hip stem_unroll)."""
    def build(w):
        co = w.shape[0]
        wu = torch.zeros((co, 7, 8, 4), dtype=torch.float32, device=w.device)
        wu[:, :, :7, :3] = w.permute(0, 2, 3, 1)  # [co, kh, kw, c]
        return _cast(wu.reshape(co, 7, 1, 32), dtype)
    return _cached(param, "stem_unrolled", dtype, build)


def dgrad_classes(k, stride, pad):
    """Per output-parity class a of the data gradient: (a, taps kh in ascending
    input offset, pad') such that dx[s*h'+a] = sum_t dy[h' - pad' + t] * W[kh_t]."""
    out = []
    for a in range(stride):
        taps = [(kh, (a + pad - kh) // stride) for kh in range(k) if (a + pad - kh) % stride == 0]
        taps.sort(key=lambda t: t[1])
        if taps:
            offs = [o for _, o in taps]
            assert offs == list(range(offs[0], offs[0] + len(offs)))
            out.append((a, [kh for kh, _ in taps], -offs[0]))
        else:
            out.append((a, [], 0))
    return out


def w_dgrad(param, dtype, khs, kws):
    """[Cout,Cin,KH,KW] -> [Cin,len(khs),len(kws),Cout] taking the listed taps in order."""
    if param.is_cuda and param.dim() == 4 and param.is_contiguous() and len(khs) <= 8 and len(kws) <= 8:
        return _form(param, dtype, ("dgrad", tuple(khs), tuple(kws)), 1, list(khs), list(kws))

    def build(w):
        w = w[:, :, khs][:, :, :, kws]
        return _cast(w.permute(1, 2, 3, 0), dtype)
    return _cached(param, ("dgrad", tuple(khs), tuple(kws)), dtype, build)


def w_linear(param, dtype, nhwc_from=None):
    """[O,I] -> [O,1,1,I]; ``nhwc_from=(C,HW)`` permutes the input index from the
    reference's NCHW flatten (c*HW+p, multilayerPerceptron.py:20) to NHWC (p*C+c)."""
    def build(w):
        o, i = w.shape
        if nhwc_from is not None:
            c, hw = nhwc_from
            w = w.reshape(o, c, hw).permute(0, 2, 1).reshape(o, i)
        return _cast(w.reshape(o, 1, 1, i), dtype)
    return _cached(param, ("lin", nhwc_from), dtype, build)


def w_linear_dgrad(param, dtype, nhwc_from=None):
    """[O,I] -> [I,1,1,O]"""
    def build(w):
        o, i = w.shape
        if nhwc_from is not None:
            c, hw = nhwc_from
            w = w.reshape(o, c, hw).permute(0, 2, 1).reshape(o, i)
        return _cast(w.t().reshape(i, 1, 1, o), dtype)
    return _cached(param, ("lin_dgrad", nhwc_from), dtype, build)


# ----------------------------------------------------------------------------
# conv + BN + act unit
# ----------------------------------------------------------------------------
class _Rec(object):
    __slots__ = ("x", "y", "out", "conv", "bn", "k", "stride", "pad", "relu", "has_res", "mean", "invstd", "scale",
                 "count", "world", "training", "form", "in_hw", "fused", "shift", "bits", "fold", "gram")


# Pointwise expanding convolutions with few input channels (conv3 / downsample of the first stages) are HBM-bound
# on their OUTPUT: they run as a fused unit that recomputes the cheap GEMM instead of storing / re-reading the raw
# conv output (forward: statistics-only pass + BN/residual/ReLU epilogue; backward: reduce + apply epilogues).
# Measured on MI355X at B=256 (bench.py, MAAI_FUSE_MAX_CIN=128): the BN/ReLU passes shrink from 161 to 82 ms but the
# extra conv passes cost 92 ms, because the short-K conv workgroups are latency-serialised (launch, load wait,
# MFMA, epilogue) rather than streaming at the HBM rate — net -3 %.  So the fused unit is OFF by default (0) until
# the pointwise kernel is persistent; the kernels and their bit-exactness tests stay in place.
_FUSE = {"max_cin": int(os.environ.get("MAAI_FUSE_MAX_CIN", "0")),
         # the same recompute trade for forward passes that keep nothing for a backward (the no-grad view of the
         # SimCLR step, evaluation): only the two forward conv passes are paid, none of the backward ones
         # Round 3: ON for the expanding pointwise units with 128 .. 256 input channels (conv3 of stages 2-3).  With the
         # streaming kernel's statistics-only and BatchNorm-epilogue launches (both take the lazy input, the second adds the
         # shortcut) the unit is a 0.3 ms + 0.8 .. 1.4 ms pair instead of launch + join (pass, or join on load in the next conv1):
         # the SimCLR step 820.0 / 822.2 -> 829.8 / 832.7 images/s (interleaved, one box).  Bit-identical to the unfused path.
         "nograd_max_cin": int(os.environ.get("MAAI_FUSE_NOGRAD_MAX_CIN", "256")),
         # ... from this many input channels up (64-channel units are the chained block boundaries of layer 1, which beat it)
         "nograd_min_cin": int(os.environ.get("MAAI_FUSE_NOGRAD_MIN_CIN", "128"))}


def _fusable(conv, form, keep=True, fold=False):
    """``fold``: the unit's backward will be the folded one (``_FOLD``), which reads neither the raw output nor recomputes it —
    the forward may then take the recompute form (statistics-only launch + BatchNorm-epilogue launch) with gradients too."""
    if not (form == "fwd" and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.out_channels >= 4 * conv.in_channels):
        return False
    if conv.in_channels <= _FUSE["max_cin"]:
        return True
    return (not keep or fold) and _FUSE["nograd_min_cin"] <= conv.in_channels <= _FUSE["nograd_max_cin"]


def _bn_training(bn):
    """does this norm layer take batch statistics in this forward?  (unit_fwd's rule)"""
    return False if _is_frozen_bn(bn) else bool(bn.training or (bn.running_mean is None))


def _is_frozen_bn(bn):
    return (not isinstance(bn, torch.nn.modules.batchnorm._BatchNorm)
            and all(hasattr(bn, a) for a in ("weight", "bias", "running_mean", "running_var")))


def _bn_eps(bn):
    return float(getattr(bn, "eps", 1e-5))


def _sync_world(bn):
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def _check_conv(conv):
    if conv.groups != 1 or conv.dilation not in ((1, 1), 1):
        raise MaaiError("HIP path supports groups=1, dilation=1 convolutions (got groups=%s dilation=%s)" % (conv.groups, conv.dilation))
    if conv.bias is not None:
        raise MaaiError("HIP path expects bias-free convolutions (resnet.py:22,28)")


_DUAL_BN = {"enabled": os.environ.get("MAAI_DUAL_BN", "1") != "0"}
# Inference with frozen statistics: conv + BN (+ shortcut) + ReLU in ONE launch per unit (MAAI_EPI_BN_ACT) where the tensors
# are small enough for launch count to matter; on large ones the lazy / streaming / chained path of the training forward
# (minus its statistics) is faster (ResNet-50, 224^2 x 256: 79.6 vs 67.8 ms).  max_rows: output pixels up to which a unit fuses.
# "fast" (round 3): beyond max_rows a unit still fuses where the epilogue runs on the kernel the plain launch would use
# (streaming, ping-pong, halo, the ring kernel's own 128-row tile: kernels.conv_bn_act_fast) AND its activation would
# otherwise be written by a BatchNorm pass — i.e. not where the consumer forms it on load for free or the block boundary
# is chained (layer 1).  ResNet-50, 224^2 x 256: 63.3 -> see DESIGN section 5.  MAAI_EVAL_FUSE_FAST=0 turns the rule off.
_EVAL_FUSE = {"enabled": os.environ.get("MAAI_EVAL_FUSE", "1") != "0", "max_rows": int(os.environ.get("MAAI_EVAL_FUSE_MAX_ROWS", "65536")),
              "fast": os.environ.get("MAAI_EVAL_FUSE_FAST", "1") != "0"}
# Normalise-on-load (kernels.Lazy): a unit whose only consumers are convolutions of this library does not run its
# BatchNorm/ReLU pass; it hands on its RAW convolution output with (scale, shift) and the consumers (forward
# convolution, weight gradient) apply the transform to the operand they stage in LDS.  "lazy": inside a block
# (conv1 -> conv2 -> conv3, and the stem -> layer1); "join": the residual join relu(bn3(y3) + shortcut) is formed by
# the NEXT block's conv1, which hands the joined activation back once for the shortcut and the backward pass.
# Bit-identical to the materialised path (tests/test_gpu_xf.py); MAAI_LAZY=0 / MAAI_JOIN=0 switch them off.
# WHERE it pays was measured per layer shape of the benchmark (scripts/xf_ab.py, B = 256, interleaved in one process;
# gain = BatchNorm pass + plain launch - lazy launch, per launch):
#   forward, one raw tensor:  stem -> layer1 +0.61 ms, layer1 conv2 (3x3 C64 @224) +0.29, layer1 conv3 +0.25,
#       layer2.0 conv2 (3x3 stride 2, C128 @224) +0.64, layer2 conv3 +0.16, layer2 conv2 @112 -0.36,
#       layer3 -0.46 / +0.02, layer4 -0.55 / -0.02  (the transform is vector work in the GEMM's K loop, repeated for
#       every re-staging of an element: it hides under HBM-bound layers and costs compute-bound ones their MFMA time)
#   residual join in the next conv1:  layer1 +0.76 (the join then runs AT the HBM rate), layer2 +0.11, layer3 -0.29, layer4 -0.72
#   weight gradient on a raw tensor (no pass saved, the transform is pure cost):  3x3 +0.65..0.81 ms, 1x1 C64 -0.05,
#       C128 +0.44, C256 +0.48
# Hence ("auto"): the consumer of a raw tensor forms it on load when it is a pointwise layer with <= 64 input channels
# (<= 128 when no backward pass will follow), or — without a backward pass only — a 3x3 layer with <= 64 channels or a
# stride-2 3x3 layer with <= 128; a block's conv1 joins on load up to 512 input channels.  "all": wherever possible
# (tests).  MAAI_LAZY_POLICY overrides.
_LAZY = {"enabled": os.environ.get("MAAI_LAZY", "1") != "0", "join": os.environ.get("MAAI_JOIN", "1") != "0",
         "policy": os.environ.get("MAAI_LAZY_POLICY", "auto")}
# Chained block boundary (csrc/conv_chain.hip): in a forward that no backward follows, the last convolution of a
# 64 -> 256 bottleneck runs as a statistics-only launch and the next block's conv1 recomputes it inside the launch that
# joins it with the shortcut — y3 is neither written nor read (measured at 224^2 x 256 images, per boundary: conv3 1.7 +
# join 4.1 ms -> statistics 0.73 + chained 3.5 ms).  With a backward pass y3 has to be stored and the chained launch
# loses (6.3 vs 5.8 ms), so those forwards keep the two launches.  MAAI_CHAIN=0 turns it off.
_CHAIN = {"enabled": os.environ.get("MAAI_CHAIN", "1") != "0"}
# Statistics of a chained conv3 from Gram(x) and colsum(x) instead of a statistics-only launch (see _unit_fwd_gen; MAAI_GRAM_STATS=0:
# the launch — whose slab is the one the unchained path sums, so that chained and unchained forwards agree bit for bit; with the
# Gram form they agree to fp32 summation order, ~1e-6 in the statistics).
# MAAI_GRAM_STATS_FUSED=0: only the chained boundaries of layer 1, not the recompute-form units of stages 2-3.
_GRAMSTATS = {"enabled": os.environ.get("MAAI_GRAM_STATS", "1") != "0", "fused": os.environ.get("MAAI_GRAM_STATS_FUSED", "1") != "0",
              # below this many pixels a statistics-only launch costs nothing, and a batch variance over a handful of pixels can sit
              # arbitrarily far below w^T Gram w / M, the scale the Gram form's fp32 rounding is relative to
              "min_rows": int(os.environ.get("MAAI_GRAM_STATS_MIN_ROWS", "1024"))}


def set_gram_stats(flag):
    _GRAMSTATS["enabled"] = bool(flag)


def set_chain(enabled):
    """Chained block boundaries in forwards without a backward pass (default on; MAAI_CHAIN=0).  Everything the chained
    launch writes — output, joined activation, mask, statistics slab — is bit-identical to the launches it replaces."""
    _CHAIN["enabled"] = bool(enabled)


def _eval_unit_fast(conv, x, dtype):
    """Inference with frozen statistics: does conv + BN (+ shortcut) + ReLU in one launch run at the plain launch's speed for
    this unit (``x``: its input, a tensor or a Lazy)?  A lazy input must be a single tensor the streaming kernel can form."""
    if not _EVAL_FUSE["fast"] or dtype != torch.bfloat16:
        return False
    lazy = isinstance(x, K.Lazy)
    if lazy and (x.b is not None or x.pre is not None):
        return False
    env = os.environ   # (a forced ring-kernel tile — the parity tests sweep those knobs — is not what the rule was measured on)
    if any(k in env for k in ("MAAI_CONV_BM", "MAAI_CONV_BN", "MAAI_CONV_NSTAGE")):
        return False
    n, h, w = x.shape[0], x.shape[1], x.shape[2]
    return K.conv_bn_act_fast(conv, n, h, w, dtype, lazy=lazy)


def _chain_ok(conv, x, dtype):
    """Can ``conv`` (the last convolution of a block, input ``x``) be recomputed by the next block's chained launch?"""
    if not (_CHAIN["enabled"] and dtype == torch.bfloat16 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.in_channels == 64 and conv.out_channels == 256):
        return False
    if isinstance(x, K.Lazy) and (x.b is not None or x.pre is not None):
        return False
    # the statistics-only launch of a lazy input exists on the streaming kernel only: not when it is switched off or a
    # ring-kernel tile is forced (the parity tests do both)
    env = os.environ
    return env.get("MAAI_CONV_PWS", "1") != "0" and not any(k in env for k in ("MAAI_CONV_BM", "MAAI_CONV_BN", "MAAI_CONV_NSTAGE"))


def set_lazy(lazy=None, join=None, policy=None):
    if lazy is not None:
        _LAZY["enabled"] = bool(lazy)
    if join is not None:
        _LAZY["join"] = bool(join)
    if policy is not None:
        if policy not in ("auto", "all"):
            raise ValueError("lazy policy must be 'auto' or 'all'")
        _LAZY["policy"] = policy


def _lazy_pays(consumers, keep):
    """Should a unit hand its raw output to ``consumers`` (convolutions) instead of running its BatchNorm pass?"""
    if not _LAZY["enabled"]:
        return False
    if _LAZY["policy"] == "all":
        return True
    for conv in consumers:
        k, cin, stride = conv.kernel_size[0], conv.in_channels, conv.stride[0]
        if k == 1:
            # (the streaming kernel normalises once per element: free for the expanding layers it takes, up to 256 channels;
            #  with a backward pass the weight gradient would pay for it instead)
            ok = cin <= (64 if keep else 128) or (not keep and cin <= 256 and stride == 1 and conv.out_channels >= 2 * cin)
        elif k == 3:
            ok = (not keep) and (cin <= 64 or (stride == 2 and cin <= 128))
        else:
            ok = False
        if not ok:
            return False
    return True


def materialise(x):
    """The tensor a ``kernels.Lazy`` activation stands for (one BatchNorm pass); tensors pass through."""
    if not isinstance(x, K.Lazy):
        return x
    y = x.y
    if x.pre is not None:   # a tensor that was never stored: one plain launch of its producer
        y = K.conv2d(x.pre[0], x.pre[1])
    if x.b is None:
        return K.bn_act_fwd(y, x.scale, x.shift, None, x.relu)
    if x.scale2 is None:
        return K.bn_act_fwd(y, x.scale, x.shift, x.b, x.relu)
    return K.bn_act_fwd2(y, x.scale, x.shift, x.b, x.scale2, x.shift2, x.relu)


def unit_output(rec):
    """The activation a unit's record stands for (a lazy unit never stored it: one BatchNorm pass rebuilds it)."""
    if rec.out is not None:
        return rec.out
    if rec.has_res:
        raise MaaiError("unit_output: the join of this unit has not been formed yet")
    return K.bn_act_fwd(rec.y, rec.scale, rec.shift, None, rec.relu)


def _lazy_input_ok(x, conv, dtype):
    """Can ``conv`` form the lazy activation ``x`` on load?  (a two-tensor join needs a pointwise stride-1 layer)"""
    cin = conv.in_channels
    if cin % (32 if dtype == torch.bfloat16 else 16) or cin > 4096:
        return False
    if x.b is not None:
        return conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and cin <= 2048
    return True


# SyncBatchNorm exchanges travel in fp32 (2C+1 / 2C words, <= 16 KB: latency-bound messages).  A unit generator yields
# ("gather", packed) — its mean | M2 | count row (kernels.bn_pack_stats) — in the forward pass and resumes with the
# [world, 2C+1] matrix of every rank's row, or ("reduce", sums) — fp32 sums of the BatchNorm backward — and resumes with
# their all-reduced values.  EXCHANGES counts the collectives issued (the per-step latency budget).
EXCHANGES = {"gather": 0, "reduce": 0}


def _exchange(kind, vecs):
    """ONE collective for all of ``vecs`` (same kind): returns the per-vector results, in order."""
    flat = vecs[0] if len(vecs) == 1 else torch.cat(vecs)
    EXCHANGES[kind] += 1
    if kind == "gather":
        world = dist.get_world_size()
        n = flat.numel()
        buf = torch.empty(world * n, dtype=flat.dtype, device=flat.device)   # (flat: what every backend's gather accepts)
        p2p = None
        if flat.is_cuda and os.environ.get("MAAI_P2P_GATHER", "0") == "1":
            # the 98 forward gathers of a step are latency-bound messages of <= 16 KB on the compute stream: with
            # MAAI_P2P_GATHER=1 they take the one-kernel direct all-gather of csrc/comm.hip (adopted by all ranks or none;
            # consecutive gathers synchronise themselves: nobody finishes epoch e before everybody has written it, so no rank
            # runs more than one gather ahead — the two epoch parities suffice)
            from . import comm
            p2p = comm.p2p_gather_for(4 * n, tag="syncbn")
        if p2p is not None:
            p2p.gather(flat.contiguous().view(1, n), buf.view(world, n))
            EXCHANGES["p2p"] = EXCHANGES.get("p2p", 0) + 1
        else:
            dist.all_gather_into_tensor(buf, flat.contiguous())
        out = buf.view(world, n)
        res, o = [], 0
        for v in vecs:
            res.append(out[:, o:o + v.numel()])
            o += v.numel()
        return res
    if kind != "reduce":
        raise MaaiError("unknown exchange %r" % (kind,))
    dist.all_reduce(flat)     # (the vectors are fresh fp32 copies of the units' fp64 sums: nothing of a caller's is reduced in place)
    res, o = [], 0
    for v in vecs:
        res.append(flat[o:o + v.numel()])
        o += v.numel()
    return res


# nn.BatchNorm2d.num_batches_tracked += 1 (torch/nn/modules/batchnorm.py: every training forward).  Inside backbone_fwd the
# increments of a whole forward are collected and applied by ONE torch._foreach_add_ (53 tiny launches -> 1 per forward);
# a unit driven on its own (tests, the head) is incremented at once.
_NBT = {"depth": 0, "pending": []}


def _count_batch(bn):
    if _DEFER["stats"] is not None:
        _DEFER["nbt"].append(bn.num_batches_tracked)    # (applied with the running statistics, after the streams have joined)
    elif _NBT["depth"] > 0:
        _NBT["pending"].append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


# The two forwards of a SimCLR step on two HIP streams (MAAI_OVERLAP_VIEWS=1 / set_overlap_views): the
# no-grad view-1 forward (Contrastive_Learning.py:638-640 under torch.no_grad, as bench.py runs it) is enqueued on a side
# stream and the gradient-carrying view-2 forward on the caller's stream right behind it; the device then has launches of
# both to overlap (an HBM-bound launch of one next to an MFMA-bound one of the other, tails next to heads).  The two are
# independent but for the BatchNorm buffers: while a forward is in flight on the side stream, every training-mode BatchNorm
# of EITHER forward writes its (float)mean / (float)unbiased variance to scratch instead of touching running_mean / running_var
# / num_batches_tracked, and when the second forward has been enqueued the caller's stream waits for the side stream and ONE
# launch applies both updates of every layer in program order (kernels.bn_running_update_multi): the buffers are bit-identical
# to two forwards run one after the other.  The output of the side forward is valid on the caller's stream after that join —
# which also happens at the next loss / optimiser / forward call, whichever comes first (``flush_overlap``).
# Single rank only (SyncBatchNorm's collectives stay on one stream), training mode, no block recompute.
_OVL = {"enabled": os.environ.get("MAAI_OVERLAP_VIEWS", "0") == "1", "stream": None, "pending": None}
_DEFER = {"stats": None, "nbt": None, "buf": None, "bufs": None}


def set_overlap_views(flag):
    flush_overlap()
    _OVL["enabled"] = bool(flag)


class _Pending(object):
    __slots__ = ("stream", "stats", "nbt", "out", "bufs")


def _defer_stats(bn, c, device):
    """Scratch for one BatchNorm's deferred statistics: (mean, unbiased variance) fp32 [C] each, zeroed — slices of ONE flat
    buffer per forward (``_DEFER["buf"]``: [tensor, offset], grown in 64 K-float steps); recorded for the join."""
    buf = _DEFER["buf"]
    if buf[0] is None or buf[1] + 2 * c > buf[0].numel():
        buf[0], buf[1] = torch.zeros(max(65536, 2 * c), dtype=torch.float32, device=device), 0
        _DEFER["bufs"].append(buf[0])
    m, v = buf[0][buf[1]:buf[1] + c], buf[0][buf[1] + c:buf[1] + 2 * c]
    buf[1] += 2 * c
    _DEFER["stats"].append((bn, m, v))
    return m, v


def _defer_begin(stats, nbt):
    _DEFER["stats"], _DEFER["nbt"], _DEFER["buf"], _DEFER["bufs"] = stats, nbt, [None, 0], []


def _defer_end():
    bufs = _DEFER["bufs"]
    _DEFER["stats"] = _DEFER["nbt"] = _DEFER["buf"] = _DEFER["bufs"] = None
    return bufs


def flush_overlap():
    """Join a forward that is in flight on the side stream (if any) and apply the deferred BatchNorm buffer updates."""
    pend = _OVL["pending"]
    if pend is None:
        return
    _OVL["pending"] = None
    _join_overlap(pend, None, None)


def _join_overlap(pend, stats2, nbt2):
    cur = torch.cuda.current_stream()
    cur.wait_stream(pend.stream)
    for t in ([pend.out] if pend.out is not None else []) + list(pend.bufs or []):
        t.record_stream(cur)   # (allocated on the side stream, read on this one from here on)
    order, by = [], {}
    for lst in (pend.stats, stats2 or []):
        for (bn, m, v) in lst:
            e = by.get(id(bn))
            if e is None:
                e = by[id(bn)] = [bn, [], []]
                order.append(e)
            e[1].append(m)
            e[2].append(v)
    items = []
    for bn, ms, vs in order:
        if len(ms) > 2:
            raise MaaiError("overlapped forwards: a BatchNorm layer ran more than twice between two joins")
        mom = bn.momentum
        items.append((bn.running_mean, ms[0], ms[1] if len(ms) > 1 else None, mom))
        items.append((bn.running_var, vs[0], vs[1] if len(vs) > 1 else None, mom))
    K.bn_running_update_multi(items)
    counts = {}
    for t in list(pend.nbt) + list(nbt2 or []):
        e = counts.setdefault(id(t), [t, 0])
        e[1] += 1
    for k in (1, 2):
        ts = [t for t, n in counts.values() if n == k]
        if ts:
            torch._foreach_add_(ts, k)
    if any(n > 2 for _, n in counts.values()):
        raise MaaiError("overlapped forwards: a BatchNorm counter was incremented more than twice between two joins")


def prewarm_weights(f, g, dtype, pool=None):
    """Every kernel-layout weight copy a forward of (f, g) reads, (re)built NOW on the current stream — before two forwards
    that both read them are enqueued on two streams."""
    for m in f.modules():
        if isinstance(m, torch.nn.Conv2d) and m is not f.conv1:
            w_fwd(m.weight, dtype)
    c1 = f.conv1
    cin = c1.weight.shape[1]
    if cin == 3 and c1.kernel_size == (7, 7) and c1.stride == (1, 1) and c1.padding == (3, 3):
        w_stem_unrolled(c1.weight, dtype)
    else:
        w_fwd(c1.weight, dtype, (cin + 31) // 32 * 32)
    if g is not None and hasattr(g, "layers"):
        l0, l2 = g.layers[0], g.layers[2]
        last = [m for m in f.modules() if isinstance(m, torch.nn.Conv2d)][-1]
        c = None
        for blk in _blocks(f):
            c = (blk.conv3 if _is_bottleneck(blk) else blk.conv2).out_channels
        if c and l0.in_features % c == 0:
            w_linear(l0.weight, dtype, (c, l0.in_features // c))
        w_linear(l2.weight, torch.float32)


class _batched_counters(object):
    def __enter__(self):
        _NBT["depth"] += 1

    def __exit__(self, et, ev, tb):
        _NBT["depth"] -= 1
        if _NBT["depth"] == 0 and _NBT["pending"]:
            pend, _NBT["pending"] = _NBT["pending"], []
            torch._foreach_add_(pend, 1)


def _drive(gen):
    """Run a unit generator to completion; it yields (kind, vector) at most once when a cross-rank exchange is due and
    continues with the exchanged result."""
    try:
        kind, vec = next(gen)
    except StopIteration as e:
        return e.value
    got = _exchange(kind, [vec])[0]
    try:
        gen.send(got)
    except StopIteration as e:
        return e.value
    raise MaaiError("a unit asked for more than one statistics exchange")


def _drive_pair(ga, gb):
    """Two units whose statistics are due at the same point (the two BatchNorms that meet at a projection shortcut,
    forward and backward): ONE collective over both vectors instead of two."""
    ra = rb = None
    ya = yb = None
    try:
        ya = next(ga)
    except StopIteration as e:
        ra = (e.value,)
    try:
        yb = next(gb)
    except StopIteration as e:
        rb = (e.value,)
    sa = sb = None
    if ya is not None and yb is not None:
        if ya[0] != yb[0]:
            raise MaaiError("paired units asked for different kinds of exchange")
        sa, sb = _exchange(ya[0], [ya[1], yb[1]])
    elif ya is not None:
        sa = _exchange(ya[0], [ya[1]])[0]
    elif yb is not None:
        sb = _exchange(yb[0], [yb[1]])[0]
    for which, g, sm in ((0, ga, sa), (1, gb, sb)):
        if sm is None:
            continue
        try:
            g.send(sm)
            raise MaaiError("a unit asked for more than one statistics exchange")
        except StopIteration as e:
            if which == 0:
                ra = (e.value,)
            else:
                rb = (e.value,)
    return ra[0], rb[0]


class _Boxed(object):
    """Generator wrapper: when the wrapped unit finishes, its output (the deferred shortcut branch) is put in ``box``,
    where the unit that applies it (created with ``branch=box``) finds it on resumption."""

    def __init__(self, gen, box):
        self.gen, self.box = gen, box

    def __next__(self):
        try:
            return next(self.gen)
        except StopIteration as e:
            self.box[0] = e.value[0]
            raise

    def send(self, v):
        try:
            return self.gen.send(v)
        except StopIteration as e:
            self.box[0] = e.value[0]
            raise


def unit_fwd(*args, **kwargs):
    """out = act(BN(conv(x)) (+ residual)): see ``_unit_fwd_gen`` (this drives it, exchanging SyncBatchNorm statistics)."""
    return _drive(_unit_fwd_gen(*args, **kwargs))


def _unit_fwd_gen(x, conv, bn, relu, residual, dtype, keep, wq=None, form="fwd", defer=False, branch=None, given=None,
                  lazy_out=False, side=None, light=False):
    """out = act(BN(conv(x)) (+ residual)); x NHWC.  Returns (out, rec or None).  A generator: with SyncBatchNorm over
    more than one rank it yields ("gather", its packed mean | M2 | count row) once and resumes with every rank's rows (``_drive``);
    ``branch`` may be a one-element list filled in before that resumption (``_drive_pair``).
    ``x`` may be a ``kernels.Lazy`` activation (formed on load).  If it is a two-tensor join, the joined activation
    (and, bf16 with gradients, its 1-bit ReLU mask) comes back in ``side["joined"]`` / ``side["bits"]``.
    ``lazy_out``: do not run this unit's BatchNorm pass — return a ``kernels.Lazy`` (rec.out stays None until the
    consumer of a join patches it in).
    ``defer``: stop after the statistics — returns ((y, scale, shift), rec) with rec.out = None, for a shortcut
    branch whose normalisation is applied by the unit it is added to; ``branch`` = such a (y2, scale2, shift2)
    triple, applied and added in this unit's single BN pass (maai_bn_act_fwd2) in place of ``residual``.
    ``given``: a record of an earlier run of this very unit (block recompute): its statistics are reused, the
    convolution runs without a statistics epilogue and the running buffers are left alone.
    ``light``: the record will only be kept for its statistics (the first forward of a recomputed block): the unit may
    then be chained into its consumer exactly as in a forward without a backward pass."""
    _check_conv(conv)
    k = conv.kernel_size[0]
    stride, pad = conv.stride[0], conv.padding[0]
    if wq is None:
        wq = w_fwd(conv.weight, dtype)
    # norm layers that are not torch BatchNorm modules but carry (weight, bias, running_mean, running_var) — the
    # FrozenBatchNorm2d the DETR backbone builds the ResNet with (detr_CLA/models/backbone.py:35-67, eps 1e-5 inside its
    # forward) — are frozen statistics whatever the module's train()/eval() flag says: folded scale / shift
    frozen = _is_frozen_bn(bn)
    training = False if frozen else (bn.training or (bn.running_mean is None))
    kh, kw = wq.shape[1], wq.shape[2]
    pad_w = pad if kw > 1 else 0
    # (a "light" forward — the first pass of a recomputed block — keeps its records for their statistics only: it takes the
    #  forms of a forward without a backward)
    # (a unit whose backward is folded keeps no raw output: with gradients it takes the recompute form as well — where the
    #  streaming kernel runs both launches, the input is a stored tensor and the BatchNorm takes batch statistics)
    fold_fwd = (_FOLD["enabled"] and _FOLD["fwd"] and keep and not light and training and dtype == torch.bfloat16 and form == "fwd"
                and not isinstance(x, K.Lazy) and K.conv_bn_act_fast(conv, x.shape[0], x.shape[1], x.shape[2], dtype, lazy=True)
                and "MAAI_CONV_PWS" not in os.environ and _FUSE["max_cin"] == 0)
    fused = _fusable(conv, form, keep and not light, fold=fold_fwd) and not defer and branch is None and given is None
    fold_fwd = fold_fwd and fused
    rows_out = (x.shape[0] * ((x.shape[1] + 2 * pad - kh) // stride + 1) * ((x.shape[2] + 2 * pad_w - kw) // stride + 1))
    eval_ok = (given is None and not training and not keep and not defer and branch is None and _EVAL_FUSE["enabled"]
               and wq.shape[0] % 64 == 0)
    eval_fused = eval_ok and rows_out <= _EVAL_FUSE["max_rows"]
    eval_lazy = False    # the fused launch also forms its (single-tensor) lazy input on load
    if (eval_ok and not eval_fused and form == "fwd" and _eval_unit_fast(conv, x, dtype)
            and not (lazy_out and (residual is None or _chain_ok(conv, x, dtype)))):
        eval_fused = True
        eval_lazy = isinstance(x, K.Lazy)
    # (a layer stays on ONE kernel family whatever form its input has: the families sum the statistics slab in different
    #  orders, and the ping-pong kernel of conv_pp.hip takes tensors only — a lazy input to one of its layers is materialised)
    # (a fused unit — statistics-only launch + BatchNorm-epilogue launch — forms a single-tensor lazy input on load in both
    #  launches where the streaming kernel takes the shape; not when a backward pass will recompute the convolution from x)
    fused_lazy = (fused and (not keep or light) and isinstance(x, K.Lazy) and x.b is None and x.pre is None and dtype == torch.bfloat16
                  and K.conv_bn_act_fast(conv, x.shape[0], x.shape[1], x.shape[2], dtype, lazy=True))
    if isinstance(x, K.Lazy) and ((fused and not fused_lazy) or (eval_fused and not eval_lazy) or form != "fwd" or not _lazy_input_ok(x, conv, dtype)
                                  or K.conv_module_family(conv, x.shape[0], x.shape[1], x.shape[2], dtype) == 2):
        x = materialise(x)
        if side is not None:
            side["joined"] = x
    join_bits = bool(keep and x.dtype == torch.bfloat16 and _DGRAD_REDUCE["enabled"] and _DGRAD_REDUCE["bits"])

    def conv_x(stats):
        """the convolution of this unit; a two-tensor lazy input is joined on load and handed back through ``side``"""
        if isinstance(x, K.Lazy) and x.b is not None:
            if x.pre is not None:   # the previous block's last convolution is recomputed inside this launch
                res = K.conv2d_chained(x, wq, stats=stats, join_bits=join_bits)
            else:
                res = K.conv2d(x, wq, stride, pad, pad_w, stats=stats, join_out=True, join_bits=join_bits)
            nret = 2 if stats else 1
            if side is None:
                raise MaaiError("unit_fwd: a joined input needs ``side`` to hand the activation back")
            side["joined"] = res[nret]
            side["bits"] = res[nret + 1] if join_bits else None
            return (res[0], res[1]) if stats else res[0]
        return K.conv2d(x, wq, stride, pad, pad_w, stats=stats)
    y = None
    gram_keep = None
    # chained block boundary: this convolution is not run here (training: only its statistics are taken) — the next block's
    # first convolution recomputes it inside the launch that joins it with the shortcut
    # (with gradients too where the unit's backward is folded: it reads neither the raw output nor recomputes it)
    fold_chain = _FOLD["enabled"] and _FOLD["fwd"] and keep and training and dtype == torch.bfloat16 and _FUSE["max_cin"] == 0
    chain = (lazy_out and (not keep or light or fold_chain) and not fused and not eval_fused and not defer and given is None and form == "fwd"
             and (residual is not None or branch is not None) and _chain_ok(conv, x, dtype))
    if given is not None:
        training = given.training
        y = conv_x(False)
        mean, invstd, scale, shift, count, world = given.mean, given.invstd, given.scale, given.shift, given.count, given.world
    elif training:
        if fused or chain:
            c = wq.shape[0]
            count = x.numel() // x.shape[-1] if not isinstance(x, K.Lazy) else x.y.numel() // x.y.shape[-1]
            if (_GRAMSTATS["enabled"] and dtype == torch.bfloat16 and wq.shape[3] in K.GRAM_CHANNELS and (chain or _GRAMSTATS["fused"])
                    and count >= _GRAMSTATS["min_rows"]
                    and not (isinstance(x, K.Lazy) and (x.b is not None or x.pre is not None))):
                # the statistics of y = x W^T from Gram(x) and colsum(x) — sum y = W sx, sum y^2 = diag(W Gram W^T) — instead of a
                # statistics-only launch that computes the whole convolution to throw it away (0.35 vs 0.78 ms at 224^2 x 256);
                # deterministic (fixed-order partials).  The backward of the folded unit needs exactly these two: kept on the record.
                g64, sx64 = K.gram_deterministic(x)
                sums = K.fold_stats(wq, g64, sx64)
                if keep and not light:
                    gram_keep = (g64.float(), sx64)
            else:
                sums = K.reduce_partials(K.conv2d_stats_only(x, wq))
        else:
            y, part = conv_x(True)
            c = y.shape[-1]
            count = y.numel() // c
            sums = K.reduce_partials(part)
        world = _sync_world(bn)
        gathered = None
        if world > 1:
            # one fp32 row per rank: mean | M2 | count, merged after the all-gather (Chan) — nn.SyncBatchNorm's protocol
            gathered = yield ("gather", K.bn_pack_stats(sums, count))
        mom = bn.momentum
        if bn.track_running_stats and bn.running_mean is not None:
            if mom is None:
                bn.num_batches_tracked += 1
                mom = 1.0 / float(bn.num_batches_tracked)
            else:
                _count_batch(bn)   # (one multi-tensor increment per forward instead of one tiny launch per layer)
            rm, rv = bn.running_mean, bn.running_var
        else:
            rm = rv = None
            mom = 0.0
        if _DEFER["stats"] is not None and rm is not None:
            if gathered is not None:
                raise MaaiError("overlapped forwards are single-rank (SyncBatchNorm exchanges stay on one stream)")
            # another forward is in flight: the statistics go to scratch (momentum 1 on zeroed buffers stores exactly
            # (float)mean / (float)unbiased variance), the buffers are updated after the join
            if bn.momentum is None:
                raise MaaiError("overlapped forwards need a fixed BatchNorm momentum")
            rm, rv = _defer_stats(bn, rm.numel(), rm.device)
            mom = 1.0
        if gathered is not None:
            # (count: the MERGED sample count as a device scalar — the ranks' batches may differ, e.g. a last batch without
            #  drop_last — which the backward's 1/N takes as is: torch.nn.SyncBatchNorm uses the summed counts in both passes)
            mean, invstd, scale, shift, count = K.bn_finalize_gathered(gathered, bn.weight, bn.bias, rm, rv, mom, bn.eps)
        else:
            mean, invstd, scale, shift = K.bn_finalize(sums, count, bn.weight, bn.bias, rm, rv, mom, bn.eps)
    else:
        scale, shift = K.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, _bn_eps(bn))
        if eval_fused:
            # inference with frozen statistics: normalise (+ shortcut) + activate in the convolution's own epilogue,
            # one launch per unit and no raw conv output in HBM (MAAI_EPI_BN_ACT on any kernel size)
            return K.conv2d_bn_act(x, wq, scale, shift, residual, relu, stride, pad, pad_w), None
        if not fused and not chain:
            y = conv_x(False)
        mean = invstd = None
        xt = x.y if isinstance(x, K.Lazy) else x
        count, world = (xt.numel() // xt.shape[-1] if fused else (y.numel() // y.shape[-1] if y is not None else 0)), 1
    if isinstance(branch, list):
        branch = branch[0]   # the shortcut branch, finalised by the same exchange (_drive_pair)
    xb = side["joined"] if (side is not None and isinstance(x, K.Lazy) and x.b is not None) else x  # what the backward reads
    if fused:
        fbits = None
        if fold_fwd and relu and residual is not None and _DGRAD_REDUCE["enabled"] and _DGRAD_REDUCE["bits"]:
            out, fbits = K.conv2d_bn_act(x, wq, scale, shift, residual, relu, want_bits=True)   # the mask the backward multiplies by
        else:
            out = K.conv2d_bn_act(x, wq, scale, shift, residual, relu)
        if not keep:
            return out, None
        r = _Rec()
        r.x, r.y, r.out, r.conv, r.bn = x, None, out, conv, bn
        r.k, r.stride, r.pad, r.relu, r.has_res = k, stride, pad, relu, residual is not None
        r.mean, r.invstd, r.scale, r.count, r.world, r.training, r.form = mean, invstd, scale, count, world, training, form
        r.in_hw, r.fused, r.shift, r.bits, r.fold, r.gram = (x.shape[1], x.shape[2]), True, shift, fbits, None, gram_keep
        return out, r
    # a residual unit's ReLU mask is kept as 1 bit per element for the backward pass (bf16): the data gradient that
    # flows into this output is masked from M*C/8 bytes instead of re-reading the output tensor
    bits = None
    want_bits = (keep and y is not None and relu and (residual is not None or branch is not None) and y.dtype == torch.bfloat16
                 and _DGRAD_REDUCE["enabled"] and _DGRAD_REDUCE["bits"])
    if defer:
        if relu or residual is not None or branch is not None:
            raise MaaiError("unit_fwd: a deferred unit is a plain conv + BN shortcut branch")
        out = None
    elif lazy_out:
        # no BatchNorm pass here: the consumer convolutions form the activation on load (y None: and recompute it)
        pre = (x, wq) if y is None else None
        if branch is not None:
            lz = K.Lazy(y, scale, shift, relu, branch[0], branch[1], branch[2], pre=pre)
        else:
            lz = K.Lazy(y, scale, shift, relu, residual, pre=pre)
        if not keep:
            return lz, None
        r = _Rec()
        r.x, r.y, r.out, r.conv, r.bn = xb, y, None, conv, bn
        r.k, r.stride, r.pad, r.relu, r.has_res = k, stride, pad, relu, (residual is not None or branch is not None)
        r.mean, r.invstd, r.scale, r.count, r.world, r.training, r.form = mean, invstd, scale, count, world, training, form
        r.in_hw = (x.shape[1], x.shape[2])
        r.fused, r.shift, r.bits, r.fold, r.gram = False, shift, None, None, gram_keep
        return lz, r
    elif branch is not None:
        if residual is not None:
            raise MaaiError("unit_fwd: residual and branch are exclusive")
        out = K.bn_act_fwd2(y, scale, shift, branch[0], branch[1], branch[2], relu, want_bits=want_bits)
        if want_bits:
            out, bits = out
    elif want_bits:
        out, bits = K.bn_act_fwd(y, scale, shift, residual, relu, want_bits=True)
    else:
        out = K.bn_act_fwd(y, scale, shift, residual, relu)
    if not keep:
        return ((y, scale, shift) if defer else out), None
    r = _Rec()
    r.x, r.y, r.out, r.conv, r.bn = xb, y, out, conv, bn
    r.k, r.stride, r.pad, r.relu, r.has_res = k, stride, pad, relu, (residual is not None or branch is not None)
    r.mean, r.invstd, r.scale, r.count, r.world, r.training, r.form = mean, invstd, scale, count, world, training, form
    r.in_hw = (x.shape[1], x.shape[2])
    r.fused, r.shift, r.bits, r.fold, r.gram = False, shift, bits, None, None
    return ((y, scale, shift) if defer else out), r


# The BatchNorm-backward reduction of a unit (sums of dz and dz*(y - mean)) rides the epilogue of the data-gradient
# convolution that PRODUCES dz, instead of a separate pass that reads dz and y back (MAAI_EPI_DGRAD_REDUCE); the
# ReLU mask of a plain conv-bn-relu unit then comes from y*scale + shift > 0, so its output is not read either.
_DGRAD_REDUCE = {"enabled": os.environ.get("MAAI_DGRAD_REDUCE", "1") != "0",
                 "bits": os.environ.get("MAAI_MASK_BITS", "1") != "0"}


def _reduce_mean(rec):
    return rec.mean if rec.training else rec.bn.running_mean


# MAAI_AXF: 0 never, 1 where measured to pay (default), 2 wherever possible.  Measured in bench.py (B = 256, per launch,
# apply pass + data gradient): layer-1 conv3 (K = 256 -> 64 columns) 5.74 -> 4.47 ms; every other pointwise layer loses
# (layer-2 conv3 2.02 -> 2.47, layer-3 conv3 0.92 -> 1.73, the expanding conv1 gradients 4.93 -> 5.23 ms): the
# register-staged operand path waits out each stage's loads and, with several column tiles, reads dz and y per tile.
_AXF = {"mode": int(os.environ.get("MAAI_AXF", "1"))}


def axf_applies(rec, dz, below, need_dx=True):
    """Should the BatchNorm-backward apply of ``rec`` be formed inside its own data-gradient launch?  (pointwise bf16
    unit whose data gradient also reduces the sums of the unit below)"""
    ok = (_AXF["mode"] > 0 and need_dx and not rec.fused and rec.y is not None and rec.k == 1 and rec.stride == 1 and rec.pad == 0
          and dz.dtype == torch.bfloat16 and below is not None and _DGRAD_REDUCE["enabled"] and below.y is not None
          and rec.conv.weight.shape[0] <= 4096)
    if ok and _AXF["mode"] == 1:
        ok = rec.conv.weight.shape[0] <= 256 and rec.conv.weight.shape[1] <= 64
    if ok and _AXF["mode"] == 3:   # experiment: every channel-reducing gradient with a single column tile
        ok = rec.conv.weight.shape[1] <= 128 and rec.conv.weight.shape[0] >= 4 * rec.conv.weight.shape[1]
    return ok


def conv_dgrad(dy, weight, k, stride, pad, in_hw, dtype, out=None, accumulate=False, relu_mask=None, below=None, axf=None,
               sum_increment=False, wq_dgrad=None, x2=None, bias=None, diag=None):
    """dx [N,IH,IW,Cin] of y = conv(x, weight[Cout,Cin,k,k]) from dy [N,OH,OW,Cout]; with ``relu_mask`` (= x,
    a post-ReLU tensor) the result is also multiplied by (x > 0) in the conv epilogue.  ``below`` = the record of
    the unit whose output x is: the mask is then that unit's, and where every pixel of dx is written exactly once
    the unit's BN-backward sums are reduced in the same epilogue.  Returns (dx, fp64 sums [2C] or None).
    ``axf = (y, k1, k2, k3, dy_out)``: ``dy`` is really dz and the operand k1*dz - k2 - k3*y is formed in the launch
    (pointwise layers with ``below``; see kernels.conv2d_store_reduce).  ``sum_increment`` (with ``accumulate``): the
    returned sums are those of what this call ADDS to ``out`` — a strided pass that leaves pixels untouched can then
    complete the sums a dense first pass produced."""
    n, cin = dy.shape[0], weight.shape[1]
    ih, iw = in_hw
    cls = dgrad_classes(k, stride, pad)
    empty = any(len(c[1]) == 0 for c in cls)
    if out is None:
        alloc = torch.zeros if (empty and not accumulate) else torch.empty
        out = alloc((n, ih, iw, cin), dtype=dy.dtype, device=dy.device)
        if accumulate:
            raise MaaiError("conv_dgrad: accumulate needs an output tensor")
    incr = bool(sum_increment and accumulate)
    # (a unit below whose BatchNorm backward is folded through its convolution wants sum(g) only: its raw output stays unread)
    sum_only = below is not None and below.has_res and _fold_static(below)
    fuse = below is not None and _DGRAD_REDUCE["enabled"] and (not empty or incr) and (below.y is not None or sum_only)
    from_y = False
    relu_mask = materialise(relu_mask)
    if below is not None:
        if not below.relu:
            raise MaaiError("conv_dgrad: the unit below has no ReLU to take a mask from")
        from_y = fuse and not below.has_res
        relu_mask = None if from_y else below.out
        if relu_mask is None and not from_y:
            # a lazy unit never stored its activation: rebuild the mask tensor (only without the fused reduction)
            if below.has_res:
                raise MaaiError("conv_dgrad: the joined activation of the unit below was never handed back")
            relu_mask = K.bn_act_fwd(below.y, below.scale, below.shift, None, True)
    use_bits = fuse and below.has_res and getattr(below, "bits", None) is not None
    if use_bits:
        relu_mask = below.bits
    launches = []
    for (a, khs, pad_h) in cls:
        for (b, kws, pad_w) in cls:
            if not khs or not kws:
                continue
            gh, gw = (ih - a + stride - 1) // stride, (iw - b + stride - 1) // stride
            if gh <= 0 or gw <= 0:
                continue
            # (``wq_dgrad``: the pointwise layer's [Cin,1,1,Cout] weights as the caller built them — the folded unit's k1*W)
            launches.append((wq_dgrad if wq_dgrad is not None else w_dgrad(weight, dtype, khs, kws), pad_h, pad_w, (gh, gw), (a, b)))
    if axf is not None and not (fuse and len(launches) == 1):
        raise MaaiError("conv_dgrad: the transformed operand needs a single fused pointwise launch")
    if not fuse:
        if x2 is not None or bias is not None or diag is not None:
            raise MaaiError("conv_dgrad: the two-source input rides the fused (DGRAD_REDUCE) launch only")
        for (wq, pad_h, pad_w, grid, off) in launches:
            K.conv2d(dy, wq, 1, pad_h, pad_w, out=out, grid_hw=grid, out_hw=(ih, iw), out_stride=stride, out_off=off,
                     accumulate=accumulate, relu_mask=relu_mask)
        return out, None
    if diag is not None and not from_y:
        raise MaaiError("conv_dgrad: the fp32 diagonal needs the unit below's activation recomputed from its raw output (a plain conv-bn-relu unit)")
    rows = [K.conv2d_stats_rows(dy, wq, 1, pad_h, pad_w, grid, (ih, iw), stride, off, axf=axf is not None, x2=x2)
            for (wq, pad_h, pad_w, grid, off) in launches]
    slab = torch.empty((sum(rows), 2, cin), dtype=torch.float32, device=dy.device)
    r0 = 0
    for (wq, pad_h, pad_w, grid, off), nr in zip(launches, rows):
        K.conv2d_store_reduce(dy, wq, 1, pad_h, pad_w, out, slab[r0:r0 + nr], None if sum_only else below.y, _reduce_mean(below),
                              below.scale if from_y else None, below.shift if from_y else None, relu_mask,
                              grid_hw=grid, out_hw=(ih, iw), out_stride=stride, out_off=off, accumulate=accumulate,
                              mask_bits=use_bits, axf=axf, sum_increment=incr, x2=x2, bias=bias, diag=diag)
        r0 += nr
    return out, K.reduce_partials(slab)


def _grad_to_reference(rec, dw):
    """kernel-layout fp32 weight gradient -> reference [Cout,Cin,KH,KW]"""
    w = rec.conv.weight
    if rec.form == "stem_unrolled":
        co = dw.shape[0]
        return dw.reshape(co, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2).contiguous()
    g = dw.permute(0, 3, 1, 2)
    if g.shape[1] != w.shape[1]:
        g = g[:, :w.shape[1]]
    return g.contiguous()


# BatchNorm-backward FOLDED through a channel-expanding pointwise convolution (conv3 of the bottlenecks, resnet.py:118-119).
# For y = x W^T (x: [M, Cin] the unit's input, W: [Cout, Cin]) followed by training-mode BatchNorm, the backward
#     dy = k1*g - k2 - k3*y,   S1 = sum g,  S2 = sum g*(y - mean)         (g = gradient of the BatchNorm's output)
# needs y only through LINEAR functions of y = x W^T, which can be moved onto the [Cout, Cin] / [Cin, Cin] side:
#     G1 = g^T x                 (a weight-gradient launch on the UN-normalised gradient: no dependence on k1..k3)
#     S2 = rowsum(W * G1) - mean * S1
#     dW = k1 * G1 - k2 (x) colsum(x) - k3 * (W Gram),   Gram = x^T x
#     dx = g (k1 * W) - x (W^T diag(k3) W) - k2 W
# so neither the BatchNorm-backward apply pass (read g, read y, write dy: three Cout-wide tensors) nor the read of y by the
# epilogue that reduces S2 exists, and y itself is not needed by the backward pass at all (4 x fewer channels in x).
# Arithmetic: exact in real numbers; in bf16 storage y is never rounded (the unfolded path rounds it once), the folded
# weights k1*W and W^T diag(k3) W are rounded to bf16 once each.  Pinned against the fp64 oracle per block
# (tests/test_gpu_fold.py) — not bit-identical to the unfolded path.  MAAI_FOLD=0 turns it off.
# MAAI_FOLD_FWD=0: the folded units' FORWARD stays a stored-output launch (only the backward changes).
# MAAI_FOLD_CAT=0: the folded data gradient as two launches (x T first, then g (k1 W) accumulating) instead of one two-source launch.
_FOLD = {"enabled": os.environ.get("MAAI_FOLD", "1") != "0", "fwd": os.environ.get("MAAI_FOLD_FWD", "1") != "0",
         "cat": os.environ.get("MAAI_FOLD_CAT", "1") != "0"}


def set_fold(flag, fwd=None):
    _FOLD["enabled"] = bool(flag)
    if fwd is not None:
        _FOLD["fwd"] = bool(fwd)


def _fold_applies(rec, dout):
    return dout is not None and dout.dtype == torch.bfloat16 and _fold_static(rec)


def _fold_static(rec):
    """Will ``rec``'s backward be the folded one?  (everything ``_fold_applies`` asks, but the gradient's dtype: the units
    that hand this unit its gradient ask, to leave its raw output unread)"""
    if not (_FOLD["enabled"] and rec is not None and rec.training and rec.form == "fwd"
            and rec.k == 1 and rec.stride == 1 and rec.pad == 0):
        return False
    lazy = isinstance(rec.x, K.Lazy)
    xt = rec.x.y if lazy else rec.x
    if xt is None or xt.dtype != torch.bfloat16:
        return False
    w = rec.conv.weight
    if not (w.shape[0] >= 2 * w.shape[1] and w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0):
        return False
    if lazy:
        # a normalise-on-load input (layer 1: conv3 of the bottlenecks, layer1.0's projection shortcut): the Gram kernel and the
        # weight gradient form it on load, the data gradient is csrc/conv_dfold.hip's (64 -> 256) or works on a materialised copy
        return rec.x.b is None and rec.x.pre is None and w.shape[1] in K.GRAM_CHANNELS
    return True


class _Fold(object):
    __slots__ = ("g1", "s1")


def _fold_sums(rec, dout, presums, dtype):
    """G1 = g^T x and the BatchNorm-backward sums [S1 | S2] (fp64) of a folded unit; G1 is kept on the record for the weight
    gradient."""
    w = rec.conv.weight
    cout, cin = w.shape[0], w.shape[1]
    f = _Fold()
    f.g1 = K.conv2d_wgrad(rec.x, dout, 1, 1, 1, 0, 0).reshape(cout, cin)
    if presums is not None:
        f.s1 = presums[:cout]
    else:
        f.s1 = K.bn_act_bwd_reduce(dout, None, None, None, False)[:cout]
    wq = w_fwd(w, dtype).reshape(cout, cin)
    s2 = K.fold_s2(wq, f.g1, f.s1, rec.mean)
    rec.fold = f
    return torch.cat([f.s1, s2])


def unit_bwd_coeffs(rec, dout, grads, dtype, presums=None):
    """BatchNorm-backward coefficients (k1, k2, k3) of a unit: see ``_unit_bwd_coeffs_gen``."""
    return _drive(_unit_bwd_coeffs_gen(rec, dout, grads, dtype, presums))


def _unit_bwd_coeffs_gen(rec, dout, grads, dtype, presums=None):
    """BatchNorm-backward coefficients (k1, k2, k3) of a unit — dy = k1*dz - k2 - k3*y — from the sums of dz and
    dz*(y - mean) (``presums`` if the producer of dz reduced them, else one reduction pass); stores the BatchNorm
    parameter gradients in ``grads``."""
    bn = rec.bn

    def reduce(mean):
        if _fold_applies(rec, dout):
            return _fold_sums(rec, dout, presums, dtype)
        if presums is not None:
            return presums
        if rec.fused:   # raw conv output never stored: recompute it inside the reduction
            return K.conv2d_bwd_reduce(rec.x, w_fwd(rec.conv.weight, dtype), dout, mean)
        return K.bn_act_bwd_reduce(dout, None, rec.y, mean, False)
    if rec.training:
        sums = reduce(rec.mean)
        gamma = bn.weight
        if rec.world > 1:
            # torch SyncBatchNorm: weight/bias gradients from the LOCAL sums, dx from the all-reduced ones
            dgamma, dbeta, _, _, _ = K.bn_bwd_coeffs(sums, rec.count, gamma, rec.mean, rec.invstd)
            sums = yield ("reduce", sums.float())   # fp32 on the wire (the local sums were taken in fp64)
            _, _, k1, k2, k3 = K.bn_bwd_coeffs(sums, rec.count, gamma, rec.mean, rec.invstd)
        else:
            dgamma, dbeta, k1, k2, k3 = K.bn_bwd_coeffs(sums, rec.count, gamma, rec.mean, rec.invstd)
    else:
        # frozen statistics: y -> y*scale + shift is a per-channel affine map
        sums = reduce(bn.running_mean)
        invstd = torch.rsqrt(bn.running_var + _bn_eps(bn))
        dbeta = sums[:sums.numel() // 2].float()
        dgamma = (sums[sums.numel() // 2:].float() * invstd)
        k1, k2, k3 = rec.scale, torch.zeros_like(rec.scale), torch.zeros_like(rec.scale)
    if bn.weight is not None and bn.weight.requires_grad:
        grads[id(bn.weight)] = dgamma
    if bn.bias is not None and bn.bias.requires_grad:
        grads[id(bn.bias)] = dbeta
    return k1, k2, k3


# The backward of a 64 -> 256 bottleneck's last unit in ONE launch (csrc/conv_bwd3.hip): BatchNorm-backward apply, data
# gradient (with the mask and the backward sums of the unit below) and weight gradient share one read of the incoming
# gradient and of y3; dz3 is never written.  Measured at 224^2 x 256 images: apply-on-load data gradient 4.0-5.7 ms +
# weight gradient 1.35 ms -> 3.75 ms.  MAAI_BWD3=0 turns it off.
_BWD3 = {"enabled": os.environ.get("MAAI_BWD3", "1") != "0"}


def _bwd3_applies(rec, dout, below, need_dx, dx_out, accumulate, relu_mask, dy):
    w = rec.conv.weight
    if _fold_applies(rec, dout):   # (a stored-tensor input: the folded backward takes the unit — it reads neither y3 nor writes dz3)
        return False
    return (_BWD3["enabled"] and dy is None and need_dx and (dx_out is None) == (not accumulate) and relu_mask is None
            and dout.dtype == torch.bfloat16 and tuple(w.shape) == (256, 64, 1, 1) and rec.stride == 1 and rec.pad == 0
            and w.requires_grad and not rec.fused and rec.y is not None and rec.form == "fwd"
            and below is not None and below.relu and not below.has_res and below.y is not None and not below.fused
            and below.scale is not None and below.shift is not None and _DGRAD_REDUCE["enabled"] and not _SIDE["enabled"])


def unit_bwd(rec, dout, grads, dtype, need_dx=True, dx_out=None, accumulate=False, relu_mask=None, below=None,
             presums=None, dy=None, sum_increment=False, coeffs=None):
    """Backward of unit_fwd.  CONVENTION: ``dout`` is already multiplied by the ReLU mask of this unit's
    output (the kernel that produced it folded ``* (out > 0)`` into its epilogue), so nothing here reads the
    forward output.  ``relu_mask`` = this unit's post-ReLU input, to pre-mask the returned dx the same way;
    ``below`` = the record of the unit that produced that input (mask AND, where possible, its BN-backward sums
    from the same epilogue).  ``presums`` = this unit's own sums if the producer of ``dout`` already reduced them;
    ``dy`` = the gradient wrt the raw conv output if the caller already ran the BatchNorm backward.
    Returns (dx or None, sums for ``below`` or None); parameter gradients go to ``grads``."""
    if _bwd3_applies(rec, dout, below, need_dx, dx_out, accumulate, relu_mask, dy):
        # (``coeffs``: the caller already has this unit's BatchNorm-backward coefficients — a projection-shortcut block
        #  finalises both of its BatchNorms in one exchange)
        k1, k2, k3 = coeffs if coeffs is not None else unit_bwd_coeffs(rec, dout, grads, dtype, presums)
        dx, slab, dw = K.conv_bwd3(dout, rec.y, below.y, w_dgrad(rec.conv.weight, dtype, [0], [0]), k1, k2, k3,
                                   _reduce_mean(below), below.scale, below.shift, dx=dx_out if accumulate else None)
        grads[id(rec.conv.weight)] = _grad_to_reference(rec, dw)
        return dx, K.reduce_partials(slab)
    if dy is None and _fold_applies(rec, dout):
        k1, k2, k3 = coeffs if coeffs is not None else unit_bwd_coeffs(rec, dout, grads, dtype, presums)
        return _unit_bwd_folded(rec, dout, grads, dtype, k1, k2, k3, need_dx, dx_out, accumulate, relu_mask, below, sum_increment)
    fused_apply = dy is None and axf_applies(rec, dout, below, need_dx) and not any(
        len(c[1]) == 0 for c in dgrad_classes(rec.k, rec.stride, rec.pad))
    if fused_apply:
        # the apply pass (dy = k1*dz - k2 - k3*y) happens inside the data-gradient launch, which also hands dy back
        # for the weight gradient: dz and y are read once, dy is written once and read once
        k1, k2, k3 = unit_bwd_coeffs(rec, dout, grads, dtype, presums)
        dy = torch.empty_like(dout) if rec.conv.weight.requires_grad else None
        dx, below_sums = conv_dgrad(dout, rec.conv.weight, rec.k, rec.stride, rec.pad, rec.in_hw, dtype, out=dx_out,
                                    accumulate=accumulate, relu_mask=relu_mask, below=below, axf=(rec.y, k1, k2, k3, dy))
        need_dx = False
    elif dy is None:
        k1, k2, k3 = unit_bwd_coeffs(rec, dout, grads, dtype, presums)
        if rec.fused:
            dy = K.conv2d_bwd_apply(rec.x, w_fwd(rec.conv.weight, dtype), dout, k1, k2, k3)
        else:
            dy, _ = K.bn_act_bwd_apply(dout, None, rec.y, k1, k2, k3, False, True, False)
    w = rec.conv.weight
    if w.requires_grad:
        kh = 7 if rec.form == "stem_unrolled" else rec.k
        kw = 1 if rec.form == "stem_unrolled" else rec.k
        pw = 0 if rec.form == "stem_unrolled" else rec.pad
        xw = rec.x
        if isinstance(xw, K.Lazy) and rec.k == 3 and xw.b is None and xw.pre is None and _lean():
            xw = materialise(xw)   # lean activations: one BatchNorm pass instead of a stored tensor (the 3x3 kernels want a tensor)
        if _SIDE["enabled"] and not K.DETAIL[0]:
            side, cur = _side_stream(), torch.cuda.current_stream()
            side.wait_stream(cur)                      # dy (and x) are ready on the main stream
            with torch.cuda.stream(side):
                dw = K.conv2d_wgrad(xw, dy, kh, kw, rec.stride, rec.pad, pw)
                grads[id(w)] = _grad_to_reference(rec, dw)
            dy.record_stream(side)                     # keep the allocator from recycling them early
            for t in ((xw.y, xw.b) if isinstance(xw, K.Lazy) else (xw,)):
                if t is not None:
                    t.record_stream(side)
            grads["_side"] = True
        else:
            dw = K.conv2d_wgrad(xw, dy, kh, kw, rec.stride, rec.pad, pw)
            grads[id(w)] = _grad_to_reference(rec, dw)
        del xw
    if not fused_apply:
        dx, below_sums = None, None
    if need_dx:
        dx, below_sums = conv_dgrad(dy, w, rec.k, rec.stride, rec.pad, rec.in_hw, dtype, out=dx_out, accumulate=accumulate,
                                    relu_mask=relu_mask, below=below, sum_increment=sum_increment)
    return dx, below_sums


_ONES = {}


def _ones(c, device):
    key = (c, device)
    if key not in _ONES:
        _ONES[key] = torch.ones(c, dtype=torch.float32, device=device)
    return _ONES[key]


def _unit_bwd_folded(rec, g, grads, dtype, k1, k2, k3, need_dx, dx_out, accumulate, relu_mask, below, sum_increment):
    """Backward of a folded unit (see ``_FOLD``): ``g`` is the gradient of the BatchNorm's OUTPUT, used as it is by the
    weight-gradient launch (already run: rec.fold.g1) and by the data gradient; y is not read."""
    f = rec.fold
    if f is None:
        raise MaaiError("unit_bwd: the folded unit's coefficients were taken without its G1")
    w = rec.conv.weight
    cout, cin = w.shape[0], w.shape[1]
    wq = w_fwd(w, dtype).reshape(cout, cin)
    x = rec.x
    if rec.gram is not None:
        gram, sx = rec.gram                    # (the forward took its statistics from them: chained boundary, _GRAMSTATS)
    elif cin in K.GRAM_CHANNELS:
        gram, sx = K.gram(x)                   # Gram = x^T x and colsum(x) in one pass over x (csrc/gram.hip)
    else:
        xm = materialise(x)
        sx = K.bn_act_bwd_reduce(xm, None, None, None, False)
        gram = K.conv2d_wgrad(xm, xm, 1, 1, 1, 0, 0).reshape(cin, cin) if w.requires_grad else None
    if w.requires_grad:
        dw = K.fold_dw(wq, f.g1, gram, sx, k1, k2, k3)
        grads[id(w)] = dw.reshape(cout, cin, 1, 1)
    rec.fold = None
    if not need_dx:
        return None, None
    xt = x.y if isinstance(x, K.Lazy) else x
    npix = xt.numel() // cin     # (the LOCAL pixel count: s1 and sx are this rank's sums)
    if (isinstance(x, K.Lazy) and (cout, cin) == (256, 64) and below is not None and below.y is x.y and below.relu and not below.has_res
            and not below.fused and x.relu and relu_mask is None and _DGRAD_REDUCE["enabled"] and (dx_out is None) == (not accumulate)):
        # layer 1: [g | a2] against the concatenated folded weights in ONE launch, a2 formed on load from the unit below's raw
        # output, that unit's mask and BatchNorm-backward sums in the epilogue (csrc/conv_dfold.hip)
        wcat, cn, dg = K.fold_dgrad_weights(wq, k1, k2, k3, f.s1, sx, npix, cat=True)
        dx, slab = K.conv_dfold(g, x.y, wcat, cn, _reduce_mean(below), below.scale, below.shift, dx=dx_out if accumulate else None, dg=dg)
        return dx, K.reduce_partials(slab)
    x = materialise(x)
    if (_FOLD["cat"] and below is not None and _DGRAD_REDUCE["enabled"] and below.y is not None and below.relu and not below.has_res
            and not below.fused and below.scale is not None and relu_mask is None and cin % 64 == 0 and cout % 64 == 0):
        # ONE launch: [g | x] against the concatenated folded weights (two-source operand of the ring / ping-pong kernels), the
        # constant added before rounding, T's diagonal in fp32 on the unit below's activation recomputed in the epilogue, that
        # unit's mask and BatchNorm-backward sums
        wcat, cn, dg = K.fold_dgrad_weights(wq, k1, k2, k3, f.s1, sx, npix, cat=True)
        return conv_dgrad(g, w, 1, 1, 0, rec.in_hw, dtype, out=dx_out, accumulate=accumulate, relu_mask=None, below=below,
                          sum_increment=False, wq_dgrad=wcat, x2=x, bias=cn, diag=dg)
    wf, tn, cn = K.fold_dgrad_weights(wq, k1, k2, k3, f.s1, sx, npix)
    # dx = g (k1 W) - x (W^T diag(k3) W) - k2 W: the short term first (K = Cin, a pointwise launch whose epilogue adds the
    # constant and — shortcut units — what is already in dx), then the long one accumulates onto it with the mask and the
    # BatchNorm-backward sums of the unit below in its epilogue
    dx0 = K.conv2d_bn_act(x, tn, _ones(cin, g.device), cn, dx_out if accumulate else None, False)
    return conv_dgrad(g, w, 1, 1, 0, rec.in_hw, dtype, out=dx0, accumulate=True, relu_mask=relu_mask, below=below,
                      sum_increment=False, wq_dgrad=wf)


def relu_mask_grad(dout, out):
    """dout * (out > 0): only needed where no conv epilogue can do it (the gradient entering the backbone)."""
    return K.bn_act_bwd_apply(dout, out, None, None, None, None, True, True, False)[0]


# ----------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------
def _blocks(resnet):
    for name in ("layer1", "layer2", "layer3", "layer4"):
        for blk in getattr(resnet, name):
            yield blk


def _block_stages(resnet):
    """stage number (1-4) of every block, in ``_blocks`` order"""
    return [i + 1 for i, name in enumerate(("layer1", "layer2", "layer3", "layer4")) for _ in getattr(resnet, name)]


def stem_input(x, conv1, dtype):
    """NCHW fp32 / list of u8 HWC views -> (stem operand NHWC, weights, form)."""
    cin = conv1.weight.shape[1]
    if isinstance(x, (list, tuple)):
        if cin != 3 * len(x):
            raise MaaiError("stem: %d views do not match conv1 with %d input channels" % (len(x), cin))
        if cin == 3:
            return K.stem_unroll(x[0], dtype), w_stem_unrolled(conv1.weight, dtype), "stem_unrolled"
        cpad = (cin + 31) // 32 * 32
        return K.pack_views_u8(list(x), cpad, dtype), w_fwd(conv1.weight, dtype, cpad), "fwd"
    if x.dim() != 4 or x.shape[1] != cin:
        raise MaaiError("stem: expected NCHW input with %d channels, got %s" % (cin, tuple(x.shape)))
    x = x.contiguous().float()
    if cin == 3 and conv1.kernel_size == (7, 7) and conv1.stride == (1, 1) and conv1.padding == (3, 3):
        return K.stem_unroll(x, dtype), w_stem_unrolled(conv1.weight, dtype), "stem_unrolled"
    cpad = (cin + 31) // 32 * 32
    return K.nchw_to_nhwc(x, cpad, dtype), w_fwd(conv1.weight, dtype, cpad), "fwd"


# Block recompute (MAAI_RECOMPUTE=1 / set_recompute): the forward keeps, per residual block, only its input and the
# per-unit BatchNorm statistics; the backward re-runs the block's forward from them (same kernels, no statistics
# epilogues, bit-identical tensors) just before differentiating it.  Activation memory drops from ~12 to ~4 C-wide
# tensors per block for one extra forward of the differentiated view.  (Round 4: 512 images per GPU — global batch 4096 on 8
# GPUs, BASELINE configs[2] — fit the device's 309 GB WITHOUT it once the activations are lean, _LEAN below: 250 GB, 915
# images/s; recompute is the memory knob beyond that.)
# Which stages: MAAI_RECOMPUTE_LAYERS (default "1,2"; bench.py --recompute asks for stage 1).  Layers 1 and 2 hold ~80 % of the
# activation bytes and cost ~60 % of a forward; keeping layers 3 and 4 stored saves their second forward (512 images / GPU,
# round 4: every stage 129 GB, stages 1-2 201 GB / 806 images/s, stage 1 242 GB / 849 images/s).
_RECOMPUTE = {"enabled": os.environ.get("MAAI_RECOMPUTE", "0") == "1",
              "layers": frozenset(int(v) for v in os.environ.get("MAAI_RECOMPUTE_LAYERS", "1,2").split(",") if v.strip())}


# Lean activations (memory): the normalised input of a bottleneck's 3x3 convolution, a1 = relu(bn1(y1)), is dropped after the
# forward launch that reads it; the backward forms it again with one BatchNorm pass (same kernel, same operands: bit-identical)
# right before the weight gradient.  y1 is stored anyway (bn1's backward reads it).  -23 GB of 141 at 256 images for ~1.5 % of
# the step: "auto" (default) = only together with block recompute, i.e. when memory is what binds (512 images per GPU);
# MAAI_LEAN_ACT=1 always, 0 never.
_LEAN = {"mode": os.environ.get("MAAI_LEAN_ACT", "auto")}


def set_lean_activations(mode):
    """"auto" | True | False (see _LEAN)."""
    _LEAN["mode"] = "auto" if mode == "auto" else ("1" if mode else "0")


def _lean():
    return _LEAN["mode"] == "1" or (_LEAN["mode"] == "auto" and _RECOMPUTE["enabled"])


def set_recompute(flag, layers=None):
    """Block recompute on / off; ``layers``: the stages (1-4) whose blocks are recomputed (default: as configured)."""
    _RECOMPUTE["enabled"] = bool(flag)
    if layers is not None:
        _RECOMPUTE["layers"] = frozenset(int(v) for v in layers)


def _lighten(r):
    if r is not None:
        r.x = r.y = r.out = r.bits = r.fold = r.gram = None
    return r


def _is_bottleneck(blk):
    return hasattr(blk, "conv3")


def _joins_on_load(blk, dtype, in_shape=None):
    """Can ``blk``'s first convolution form the previous block's residual join on load?  (``in_shape`` = (N, H, W) of the
    block input: a conv1 that runs on the ping-pong kernel takes a tensor — the join is then an ordinary BatchNorm pass)"""
    c1 = blk.conv1
    ok = (_LAZY["enabled"] and _LAZY["join"] and _is_bottleneck(blk) and c1.kernel_size == (1, 1) and c1.stride == (1, 1)
          and c1.padding == (0, 0) and c1.in_channels <= (2048 if _LAZY["policy"] == "all" else 512)
          and c1.in_channels % (32 if dtype == torch.bfloat16 else 16) == 0 and not _fusable(c1, "fwd"))
    if ok and in_shape is not None and K.conv_module_family(c1, in_shape[0], in_shape[1], in_shape[2], dtype) == 2:
        ok = False
    return ok


def _block_fwd(blk, xin, dtype, keep, given=None, lazy_out=False, pol_keep=None):
    """One residual block (resnet.py:59-77 / :113-135).  ``xin``: a tensor or a ``kernels.Lazy`` activation (the stem's,
    or the previous block's residual join).  ``lazy_out``: return the block output as a Lazy join for the next
    block's conv1 to form.  Returns (out, (r1, r2, r3, rd), xin tensor or single-tensor Lazy as materialised here,
    1-bit mask of xin or None)."""
    g1, g2, g3, gd = given if given is not None else (None, None, None, None)
    side = {}
    if pol_keep is None:
        pol_keep = keep   # does a weight gradient read this forward's activations? (not when the block is recomputed)
    light = keep and not pol_keep and given is None   # records kept for their statistics only
    needs_identity = blk.downsample is None
    if isinstance(xin, K.Lazy) and ((xin.b is None and needs_identity) or
                                    (xin.b is not None and not _joins_on_load(blk, dtype, tuple(xin.shape[:3])))):
        xin = materialise(xin)   # the identity shortcut reads it / conv1 cannot join it
    nxt = blk.conv2
    o, r1 = unit_fwd(xin, blk.conv1, blk.bn1, True, None, dtype, keep, given=g1, lazy_out=_lazy_pays([nxt], pol_keep), side=side)
    xin_bits = side.get("bits")
    if isinstance(xin, K.Lazy) and xin.b is not None:
        xin = side["joined"]     # conv1 formed the join and handed it back
    if _is_bottleneck(blk):
        a1 = o
        o, r2 = unit_fwd(o, blk.conv2, blk.bn2, True, None, dtype, keep, given=g2, lazy_out=_lazy_pays([blk.conv3], pol_keep))
        if (keep and pol_keep and given is None and r1 is not None and r2 is not None and _lean() and r2.k == 3 and r1.out is not None
                and r2.x is r1.out and a1 is r1.out and r1.y is not None and r1.relu and not r1.has_res and not r1.fused):
            r2.x = K.Lazy(r1.y, r1.scale, r1.shift, True)   # (the backward re-forms a1 from y1: unit_bwd)
            r1.out = None
        del a1
        last_conv, last_bn = blk.conv3, blk.bn3
    else:  # BasicBlock
        r2 = None
        last_conv, last_bn = blk.conv2, blk.bn2
    rd = None
    dual = blk.downsample is not None and _DUAL_BN["enabled"] and not _fusable(blk.downsample[0], "fwd") and not _fusable(last_conv, "fwd")
    if (dual and not keep and given is None and _EVAL_FUSE["enabled"] and not _bn_training(blk.downsample[1]) and not _bn_training(last_bn)
            and _eval_unit_fast(blk.downsample[0], xin, dtype) and _eval_unit_fast(last_conv, o, dtype)
            and not (lazy_out and _chain_ok(last_conv, o, dtype))):
        # frozen statistics: both branches normalise in their own epilogues, the second one adds the first (not where the
        # block boundary is chained: that launch normalises the projection shortcut on load)
        dual = False
    if dual:
        # the shortcut's BatchNorm is applied inside the last unit's pass: its normalised map is never stored; with
        # SyncBatchNorm the two units' statistics travel in ONE all-reduce (_drive_pair)
        box = [None]
        gen_d = _unit_fwd_gen(xin, blk.downsample[0], blk.downsample[1], False, None, dtype, keep, defer=True, given=gd)
        gen_3 = _unit_fwd_gen(o, last_conv, last_bn, True, None, dtype, keep, branch=box, given=g3, lazy_out=lazy_out, light=light)
        (br, rd), (out, r3) = _drive_pair(_Boxed(gen_d, box), gen_3)
    else:
        if blk.downsample is not None:
            idn, rd = unit_fwd(xin, blk.downsample[0], blk.downsample[1], False, None, dtype, keep, given=gd)
        else:
            idn = xin
        out, r3 = unit_fwd(o, last_conv, last_bn, True, idn, dtype, keep, given=g3, lazy_out=lazy_out, light=light)
    return out, (r1, r2, r3, rd), xin, xin_bits


def backbone_fwd(resnet, x, dtype, keep):
    """resnet.py:226-240.  Returns (NHWC feature map, tape)."""
    with _batched_counters():
        return _backbone_fwd(resnet, x, dtype, keep)


def _backbone_fwd(resnet, x, dtype, keep):
    xs, wq, form = stem_input(x, resnet.conv1, dtype)
    tape = []
    cin = resnet.conv1.weight.shape[1]
    blocks = list(_blocks(resnet))
    # the stem's activation is formed on load by layer1's first convolutions when that block has a projection
    # shortcut (an identity shortcut would read the tensor itself)
    ckpt_on = keep and _RECOMPUTE["enabled"] and _FUSE["max_cin"] == 0
    stages = _block_stages(resnet)
    stem_lazy = (bool(blocks) and blocks[0].downsample is not None and not _fusable(blocks[0].conv1, "fwd")
                 and not _fusable(blocks[0].downsample[0], "fwd") and _lazy_pays([blocks[0].conv1, blocks[0].downsample[0]], keep))
    K.FLOPS_SCALE[0] = (49.0 * cin) / (wq.shape[1] * wq.shape[2] * wq.shape[3])  # executed K includes zero padding
    out, r = unit_fwd(xs, resnet.conv1, resnet.bn1, True, None, dtype, keep, wq=wq, form=form, lazy_out=stem_lazy)
    K.FLOPS_SCALE[0] = 1.0
    tape.append(("stem", r))
    prev3 = None   # record of the previous block's last unit while its output is still a Lazy join
    for i, blk in enumerate(blocks):
        # (the next block's input = this block's output: its extent follows from this block's stride)
        st_i = (blk.conv2 if _is_bottleneck(blk) else blk.conv1).stride[0]
        nshape = (out.shape[0], (out.shape[1] - 1) // st_i + 1, (out.shape[2] - 1) // st_i + 1)
        lazy_out = i + 1 < len(blocks) and _joins_on_load(blocks[i + 1], dtype, nshape) and _FUSE["max_cin"] == 0
        # (block recompute: this forward's activations are dropped and rebuilt, so it runs with the no-backward policy)
        ckpt = ckpt_on and stages[i] in _RECOMPUTE["layers"]
        out, recs, xin, xin_bits = _block_fwd(blk, out, dtype, keep, lazy_out=lazy_out, pol_keep=keep and not ckpt)
        if prev3 is not None and prev3.out is None:
            prev3.out, prev3.bits = xin, xin_bits   # the join this block's conv1 formed IS the previous block's output
        prev3 = recs[2] if (lazy_out and keep) else None
        if ckpt:
            tape.append(("ckpt", blk, xin, tuple(_lighten(r) for r in recs), out if not isinstance(out, K.Lazy) else None))
        else:
            tape.append(("block",) + recs)
    return out, tape


def block_bwd(entry, dout, grads, dtype, prev=None, presums=None, mask_input=True):
    """Backward of one residual block.  ``dout`` must already carry the block output's ReLU mask; the
    returned gradient wrt the block input carries the block input's mask (folded into the last epilogue).
    ``dout`` may be overwritten (identity shortcut: the conv1 data gradient is accumulated into it).
    ``prev`` = record of the unit that produced the block input, ``presums`` = the last unit's BN-backward sums
    when the producer of ``dout`` reduced them.  ``mask_input=False`` (a block run on its own, whose input is an
    arbitrary tensor): the returned gradient is the plain d/dx.  Returns (dx, sums for ``prev`` or None)."""
    in_mask = (lambda: r1.x) if mask_input else (lambda: None)
    _, r1, r2, r3, rd = entry
    dyd, kd_fused = None, False
    if rd is not None and _DUAL_BN["enabled"] and not r3.fused and not rd.fused:
        # both branches receive the same gradient: one pass reads it once and writes both dy
        k3, kd = _drive_pair(_unit_bwd_coeffs_gen(r3, dout, grads, dtype, presums), _unit_bwd_coeffs_gen(rd, dout, grads, dtype))
        below3 = r2 if r2 is not None else r1
        if _bwd3_applies(r3, dout, below3, True, None, False, None, None):
            # the main branch's apply happens inside its fused backward launch; so does the shortcut's where that launch
            # takes it too (a stride-1 64 -> 256 projection whose input is a plain conv-bn-relu unit: ``kd_fused``)
            kd_fused = (prev is not None and rd.stride == 1 and dout.shape[1:3] == tuple(rd.in_hw)
                        and _bwd3_applies(rd, dout, prev, True, dout, True, None, None))
            if not kd_fused:
                dyd, _ = K.bn_act_bwd_apply(dout, None, rd.y, kd[0], kd[1], kd[2], False, True, False)
            d, s = unit_bwd(r3, dout, grads, dtype, below=below3, coeffs=k3)
        elif r3.fold is not None or rd.fold is not None:
            # a folded branch needs no apply pass (``_FOLD``): the other one gets the single-tensor pass
            if rd.fold is None:
                dyd, _ = K.bn_act_bwd_apply(dout, None, rd.y, kd[0], kd[1], kd[2], False, True, False)
            if r3.fold is None:
                dy3, _ = K.bn_act_bwd_apply(dout, None, r3.y, k3[0], k3[1], k3[2], False, True, False)
                d, s = unit_bwd(r3, None, grads, dtype, below=below3, dy=dy3)
                del dy3
            else:
                d, s = unit_bwd(r3, dout, grads, dtype, below=below3, coeffs=k3)
        else:
            dy3, dyd = K.bn_act_bwd_apply2(dout, r3.y, k3, rd.y, kd)
            d, s = unit_bwd(r3, None, grads, dtype, below=below3, dy=dy3)
            del dy3
    else:
        d, s = unit_bwd(r3, dout, grads, dtype, below=r2 if r2 is not None else r1, presums=presums)
    if r2 is not None:
        d, s = unit_bwd(r2, d, grads, dtype, below=r1, presums=s)
    kdc = kd if (rd is not None and rd.fold is not None) else None   # a folded shortcut unit: its coefficients are already there
    if rd is not None:
        # dx = dgrad(conv1) (dense) then += dgrad(downsample) (strided scatter, accumulate + mask epilogue)
        # (mask both: pixels the strided scatter never touches keep the first, already masked, value;
        #  m*(m*a + b) == m*(a + b) for a 0/1 mask).  The sums for ``prev`` can only ride the second pass, and
        #  only if it rewrites every pixel (stride-1 downsample); otherwise ``prev`` reduces them itself.
        strided = rd.stride != 1
        if prev is not None and strided and _DGRAD_REDUCE["enabled"] and (prev.y is not None or (prev.has_res and _fold_static(prev))):
            # stride-2 shortcut: the dense conv1 pass reduces the sums of what it stores, the strided pass those of what
            # it adds on the pixels it touches (sum_increment) — together the sums of the final gradient
            dx, sa = unit_bwd(r1, d, grads, dtype, below=prev, presums=s)
            dx, sb = unit_bwd(rd, dout, grads, dtype, dx_out=dx, accumulate=True, below=prev, dy=dyd, sum_increment=True, coeffs=kdc)
            sp = sa + sb if (sa is not None and sb is not None) else None
            if sp is None and (sa is not None or sb is not None):
                raise MaaiError("block_bwd: the two passes of a strided shortcut must both reduce or both not")
        else:
            # (a stride-1 shortcut's pass rewrites — and masks — every pixel: the first pass then needs no mask)
            dx, _ = unit_bwd(r1, d, grads, dtype, relu_mask=in_mask() if strided else None, presums=s)
            if prev is not None and kd_fused:
                dx, sp = unit_bwd(rd, dout, grads, dtype, dx_out=dx, accumulate=True, below=prev, coeffs=kd)
            elif prev is not None:
                dx, sp = unit_bwd(rd, dout, grads, dtype, dx_out=dx, accumulate=True, below=prev, dy=dyd, coeffs=kdc)
            else:
                dx, sp = unit_bwd(rd, dout, grads, dtype, dx_out=dx, accumulate=True, relu_mask=in_mask(), dy=dyd, coeffs=kdc)
    else:
        # identity shortcut: dx = dout + dgrad(conv1), accumulated in place in the conv epilogue
        if prev is not None:
            dx, sp = unit_bwd(r1, d, grads, dtype, dx_out=dout, accumulate=True, below=prev, presums=s)
        else:
            dx, sp = unit_bwd(r1, d, grads, dtype, dx_out=dout, accumulate=True, relu_mask=in_mask(), presums=s)
    return dx, sp


# N > 1: an object with ready(grads) / finish(grads) (maai_hip.dist.GradReducer).  The hand-written backward calls
# ready() after the head and after every block, so a bucket's all-reduce starts on the side stream while the blocks
# below are still being differentiated.
_GRAD_HOOK = [None]
# The reference builds DistributedDataParallel and immediately unwraps it (Contrastive_Learning.py:418-424): its replicas never
# exchange gradients and drift apart.  Under the UNCHANGED driver this engine therefore installs the
# gradient all-reduce by itself: at the first backward that finds an initialised process group of more than one rank (and no
# hook set by hand) it builds a ``dist.GradReducer`` over the parameters of that autograd node and keeps it for them.
# MAAI_GRAD_ALLREDUCE = "auto" (default: as described) | "0" (never: the reference's behaviour, replicas drift) | "1" (same as
# auto, but a missing process group at the first backward of a multi-rank launch — WORLD_SIZE > 1 — is an error).
_AUTO_REDUCE = {"mode": os.environ.get("MAAI_GRAD_ALLREDUCE", "auto"), "hooks": {}}   # hooks: {ids of the parameters: (reducer, weak refs)}


def set_grad_hook(hook):
    """An object with ready(grads) / finish(grads) (``maai_hip.dist.GradReducer``), or None: back to the automatic rule."""
    _GRAD_HOOK[0] = hook


def set_grad_allreduce(mode):
    """"auto" | "0" | "1" — see MAAI_GRAD_ALLREDUCE above."""
    if str(mode) not in ("auto", "0", "1"):
        raise ValueError("grad all-reduce mode must be 'auto', '0' or '1'")
    _AUTO_REDUCE["mode"] = str(mode)
    _AUTO_REDUCE["hooks"].clear()


def _grad_hook_for(params):
    """The gradient hook of a backward over ``params``: the one set by hand, else the automatic GradReducer (or None)."""
    if _GRAD_HOOK[0] is not None:
        return _GRAD_HOOK[0]
    mode = _AUTO_REDUCE["mode"]
    if mode == "0":
        return None
    if not (dist.is_available() and dist.is_initialized()):
        if mode == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise MaaiError("MAAI_GRAD_ALLREDUCE=1 under WORLD_SIZE > 1, but torch.distributed is not initialised at the first backward")
        return None
    if dist.get_world_size() < 2:
        return None
    key = tuple(id(p) for p in params)
    hooks = _AUTO_REDUCE["hooks"]
    hit = hooks.get(key)
    if hit is None or any(r() is not p for r, p in zip(hit[1], params)):   # (a recycled id belongs to another parameter)
        from .dist import GradReducer
        if len(hooks) >= 4:
            hooks.clear()
        hit = hooks[key] = (GradReducer(list(params)), [weakref.ref(p) for p in params])
    return hit[0]


def backbone_bwd(tape, dout, grads, dtype, hook=None):
    """``dout``: gradient wrt the layer4 map, NOT yet masked.  ``hook``: the gradient hook of this backward (``_grad_hook_for``)."""
    last = tape[-1]
    dout = relu_mask_grad(dout, last[4] if last[0] == "ckpt" else (last[3].out if last[0] == "block" else last[1].out))
    sums = None
    for i in range(len(tape) - 1, -1, -1):
        entry = tape[i]
        if entry[0] == "stem":
            r = entry[1]
            K.FLOPS_SCALE[0] = (49.0 * r.conv.weight.shape[1]) / ((7 if r.form == "stem_unrolled" else 49) * r.x.shape[-1])
            unit_bwd(r, dout, grads, dtype, need_dx=False, presums=sums)
            K.FLOPS_SCALE[0] = 1.0
            return
        if entry[0] == "ckpt":
            # rebuild this block's records from its input and the saved statistics, differentiate, drop them; the
            # block below is not materialised yet, so its BN-backward sums cannot ride this block's last epilogue
            _, blk, xin, lights, _ = entry
            # (lazy_out: the rebuilt block's OUTPUT is not needed again — the gradient arriving here is already masked and
            #  the block above was differentiated from its own stored input — so its last BatchNorm pass is not re-run)
            _, recs, _, _ = _block_fwd(blk, xin, dtype, True, given=lights, lazy_out=True)
            dout, sums = block_bwd(("block",) + recs, dout, grads, dtype, prev=None, presums=sums)
            tape[i] = None
            del recs
            if hook is not None:
                hook.ready(grads)
            continue
        below = tape[i - 1]
        prev = below[1] if below[0] == "stem" else (below[3] if below[0] == "block" else None)
        dout, sums = block_bwd(entry, dout, grads, dtype, prev=prev, presums=sums)
        if hook is not None:
            hook.ready(grads)


# ----------------------------------------------------------------------------
# head (multilayerPerceptron.py:9-22) on NHWC features
# ----------------------------------------------------------------------------
def _check_mlp(mlp):
    l0, l2 = mlp.layers[0], mlp.layers[2]
    if l0.in_features % 32 or l0.out_features % 64 or l2.out_features % 64 or l2.in_features % 32:
        raise MaaiError("HIP MLP needs in %% 32 == 0 and hidden/out %% 64 == 0 (got %d,%d,%d)" %
                        (l0.in_features, l0.out_features, l2.out_features))
    return l0, l2


def head_fwd(mlp, feat, dtype, keep, pool=None, nhwc=True):
    """feat: NHWC [B,h,w,C] (nhwc=True, W1 columns permuted on the fly) or an
    already NCHW-flattened [B,1,1,D] tensor (nhwc=False)."""
    l0, l2 = _check_mlp(mlp)
    b = feat.shape[0]
    pooled_from = None
    if nhwc:
        h, w, c = feat.shape[1:]
        if pool is not None and (h != pool or w != pool):
            pooled_from = (h, w)
            feat = K.avgpool_fwd(feat, pool, pool)
            h = w = pool
        if h * w * c != l0.in_features:
            raise MaaiError("MLP expects %d input features, backbone gives %d x %d x %d" % (l0.in_features, c, h, w))
        v = feat.reshape(b, 1, 1, h * w * c)
        perm = (c, h * w)
    else:
        v, perm = feat, None
    y1 = K.conv2d(v, w_linear(l0.weight, dtype, perm), 1, 0, 0)
    hdn = K.bn_act_fwd(y1, None, l0.bias, None, True)
    h32 = K.cast_to_f32(hdn)
    y2 = K.conv2d(h32, w_linear(l2.weight, torch.float32), 1, 0, 0)
    z = K.bn_act_fwd(y2, None, l2.bias, None, False).reshape(b, l2.out_features)
    tape = (v, hdn, h32, perm, pooled_from, feat.shape) if keep else None
    return z, tape


def head_bwd(mlp, tape, dz, grads, dtype, need_dfeat=True):
    l0, l2 = mlp.layers[0], mlp.layers[2]
    v, hdn, h32, perm, pooled_from, fshape = tape
    b = dz.shape[0]
    dz4 = dz.contiguous().float().reshape(b, 1, 1, l2.out_features)
    if l2.bias is not None and l2.bias.requires_grad:
        grads[id(l2.bias)] = K.bn_act_bwd_reduce(dz4, None, None, None, False)[:l2.out_features].float()
    if l2.weight.requires_grad:
        grads[id(l2.weight)] = K.conv2d_wgrad(h32, dz4, 1, 1).reshape(l2.out_features, l2.in_features)
    dh32 = K.conv2d(dz4, w_linear_dgrad(l2.weight, torch.float32), 1, 0, 0)
    dh = K.cast_from_f32(dh32, dtype)
    dy1, _ = K.bn_act_bwd_apply(dh, hdn, None, None, None, None, True, True, False)
    if l0.bias is not None and l0.bias.requires_grad:
        grads[id(l0.bias)] = K.bn_act_bwd_reduce(dy1, None, None, None, False)[:l0.out_features].float()
    if l0.weight.requires_grad:
        dw = K.conv2d_wgrad(v, dy1, 1, 1).reshape(l0.out_features, l0.in_features)
        if perm is not None:
            c, hw = perm
            dw = dw.reshape(l0.out_features, hw, c).permute(0, 2, 1).reshape(l0.out_features, l0.in_features).contiguous()
        grads[id(l0.weight)] = dw
    if not need_dfeat:
        return None
    dv = K.conv2d(dy1, w_linear_dgrad(l0.weight, dtype, perm), 1, 0, 0)
    if perm is None:
        return dv
    dfeat = dv.reshape(fshape)
    if pooled_from is not None:
        dfeat = K.avgpool_bwd(dfeat, pooled_from[0], pooled_from[1])
    return dfeat


# ----------------------------------------------------------------------------
# autograd nodes
# ----------------------------------------------------------------------------
def trainable_params(*modules):
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def _need_gpu_module(m):
    p = next(m.parameters(), None)
    if p is not None and not p.is_cuda:
        raise MaaiError("the HIP path needs the module on a HIP device (got %s); there is no CPU fallback — "
                        "call .to('cuda') first" % p.device)


class _FusedFn(torch.autograd.Function):
    """z = g(pool(f(x))) entirely in NHWC (SimCLR.py:25-29)."""

    @staticmethod
    def forward(ctx, x, f, g, pool, keep, *params):
        dtype = compute_dtype()
        feat, tape = backbone_fwd(f, x, dtype, keep)
        z, htape = head_fwd(g, feat, dtype, keep, pool, nhwc=True)
        ctx.f, ctx.g, ctx.tape, ctx.htape, ctx.dtype, ctx.params = f, g, tape, htape, dtype, params
        ctx.keep = keep
        return z

    @staticmethod
    def backward(ctx, dz):
        if not ctx.keep:
            raise MaaiError("backward through a forward that ran without gradients")
        grads = {}
        hook = _grad_hook_for(ctx.params)
        dfeat = head_bwd(ctx.g, ctx.htape, dz, grads, ctx.dtype)
        if hook is not None:
            hook.ready(grads)
        backbone_bwd(ctx.tape, dfeat, grads, ctx.dtype, hook)
        if grads.pop("_side", False):
            torch.cuda.current_stream().wait_stream(_side_stream())
        if hook is not None:
            hook.finish(grads)   # the compute stream continues behind the last bucket; the tensors below hold the averages
        ctx.tape = ctx.htape = None
        return (None, None, None, None, None) + tuple(grads.get(id(p)) for p in ctx.params)


class _BackboneFn(torch.autograd.Function):
    """f(x): NCHW fp32 in, NCHW fp32 out — the reference's ResNet.forward contract."""

    @staticmethod
    def forward(ctx, x, f, keep, *params):
        dtype = compute_dtype()
        feat, tape = backbone_fwd(f, x, dtype, keep)
        ctx.tape, ctx.dtype, ctx.params, ctx.keep, ctx.cpad = tape, dtype, params, keep, feat.shape[-1]
        return K.nhwc_to_nchw(feat, feat.shape[-1])

    @staticmethod
    def backward(ctx, dfeat):
        if not ctx.keep:
            raise MaaiError("backward through a forward that ran without gradients")
        grads = {}
        hook = _grad_hook_for(ctx.params)
        d = K.nchw_to_nhwc(dfeat.contiguous().float(), ctx.cpad, ctx.dtype)
        backbone_bwd(ctx.tape, d, grads, ctx.dtype, hook)
        if grads.pop("_side", False):
            torch.cuda.current_stream().wait_stream(_side_stream())
        if hook is not None:
            hook.finish(grads)
        ctx.tape = None
        return (None, None, None) + tuple(grads.get(id(p)) for p in ctx.params)


class _BlockFn(torch.autograd.Function):
    """One residual block on its own (consumers that walk ``layerN`` sub-modules, e.g. an intermediate-layer getter
    over the backbone): NCHW fp32 in and out, the reference's BasicBlock / Bottleneck.forward contract
    (resnet.py:59-77, 113-135)."""

    @staticmethod
    def forward(ctx, x, blk, keep, need_dx, *params):
        dtype = compute_dtype()
        b, c, h, w = x.shape
        xin = K.nchw_to_nhwc(x.contiguous().float(), c, dtype)
        out, recs, _, _ = _block_fwd(blk, xin, dtype, keep)
        ctx.recs, ctx.dtype, ctx.params, ctx.keep, ctx.need_dx, ctx.cin = recs, dtype, params, keep, need_dx, c
        return K.nhwc_to_nchw(out, out.shape[-1])

    @staticmethod
    def backward(ctx, dout):
        if not ctx.keep:
            raise MaaiError("backward through a forward that ran without gradients")
        grads = {}
        r3 = ctx.recs[2]
        d = K.nchw_to_nhwc(dout.contiguous().float(), dout.shape[1], ctx.dtype)
        d = relu_mask_grad(d, r3.out)
        dx, _ = block_bwd(("block",) + ctx.recs, d, grads, ctx.dtype, prev=None, mask_input=False)
        ctx.recs = None
        gx = K.nhwc_to_nchw(dx, ctx.cin) if ctx.need_dx else None
        return (gx, None, None, None) + tuple(grads.get(id(p)) for p in ctx.params)


def block_forward(blk, x):
    """``blk(x)`` for one BasicBlock / Bottleneck of this package's ResNet on the HIP engine."""
    flush_overlap()
    _need_gpu_module(blk)
    if not x.is_cuda:
        raise MaaiError("the HIP path needs tensors on a HIP device (got %s); there is no CPU fallback" % x.device)
    if x.dim() != 4 or x.shape[1] != blk.conv1.in_channels:
        raise MaaiError("block: expected NCHW input with %d channels, got %s" % (blk.conv1.in_channels, tuple(x.shape)))
    params = trainable_params(blk)
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    return _BlockFn.apply(x, blk, keep, x.requires_grad, *params)


class _HeadFn(torch.autograd.Function):
    """g(v): [B, ...] fp32 flattened in the caller's (NCHW) order."""

    @staticmethod
    def forward(ctx, v, g, keep, *params):
        dtype = compute_dtype()
        b = v.shape[0]
        flat = v.reshape(b, -1).contiguous().float()
        vin = K.cast_from_f32(flat, dtype).reshape(b, 1, 1, flat.shape[1])
        z, tape = head_fwd(g, vin, dtype, keep, None, nhwc=False)
        ctx.g, ctx.tape, ctx.dtype, ctx.params, ctx.keep, ctx.vshape, ctx.need_dv = g, tape, dtype, params, keep, v.shape, v.requires_grad
        return z

    @staticmethod
    def backward(ctx, dz):
        grads = {}
        dv = head_bwd(ctx.g, ctx.tape, dz, grads, ctx.dtype, need_dfeat=ctx.need_dv)
        ctx.tape = None
        if dv is not None:
            dv = K.cast_to_f32(dv).reshape(ctx.vshape)
        return (dv, None, None) + tuple(grads.get(id(p)) for p in ctx.params)


def _overlap_ok(f, x, keep):
    if not (_OVL["enabled"] and not keep and f.training and not _RECOMPUTE["enabled"]):
        return False
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return False
    xs = x if isinstance(x, (list, tuple)) else [x]
    return all(torch.is_tensor(t) and t.is_cuda for t in xs)


def fused_forward(f, g, x, pool=None):
    _need_gpu_module(f)
    params = trainable_params(f, g)
    keep = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    pend = _OVL["pending"]
    if pend is None and _overlap_ok(f, x, keep):
        # the no-grad view: enqueued on the side stream; the next forward (or loss / optimiser call) joins it
        if _OVL["stream"] is None:
            _OVL["stream"] = torch.cuda.Stream()
        side, cur = _OVL["stream"], torch.cuda.current_stream()
        prewarm_weights(f, g, compute_dtype(), pool)
        side.wait_stream(cur)
        pend = _Pending()
        pend.stream, pend.stats, pend.nbt, pend.out, pend.bufs = side, [], [], None, None
        _defer_begin(pend.stats, pend.nbt)
        try:
            with torch.cuda.stream(side):
                pend.out = _FusedFn.apply(x, f, g, pool, False, *params)
        finally:
            pend.bufs = _defer_end()
        for t in (x if isinstance(x, (list, tuple)) else [x]):
            t.record_stream(side)
        _OVL["pending"] = pend
        return pend.out
    if pend is None:
        return _FusedFn.apply(x, f, g, pool, keep, *params)
    # a forward is in flight on the side stream: this one defers its BatchNorm buffer updates too, then joins
    _OVL["pending"] = None
    stats2, nbt2 = [], []
    _defer_begin(stats2, nbt2)
    try:
        z = _FusedFn.apply(x, f, g, pool, keep, *params)
    finally:
        _defer_end()
        _join_overlap(pend, stats2, nbt2)
    return z


def backbone_forward(f, x):
    flush_overlap()
    _need_gpu_module(f)
    params = trainable_params(f)
    keep = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _BackboneFn.apply(x, f, keep, *params)


def head_forward(g, v):
    flush_overlap()
    _need_gpu_module(g)
    params = trainable_params(g)
    keep = torch.is_grad_enabled() and (v.requires_grad or any(p.requires_grad for p in params))
    return _HeadFn.apply(v, g, keep, *params)
