"""Tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw
device pointers + the current HIP stream.  Internal layout: activations NHWC
``[N,H,W,C]`` contiguous, weights ``[Cout,KH,KW,Cin]``; storage dtype
torch.bfloat16 (production) or torch.float32 (exact-fp32 parity mode).

No fallback: every wrapper raises ``MaaiError`` for non-HIP tensors.
"""
import ctypes as C
import os

import torch

from ._lib import (BF16, EPI_BN_ACT, EPI_BWD_APPLY, EPI_BWD_REDUCE, EPI_DGRAD_REDUCE, EPI_STATS_ONLY, EPI_STORE, F32, ConvDesc, ConvEpilogue, MaaiError,
                   check, lib)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise MaaiError("unsupported storage dtype %s (bf16 or f32)" % t.dtype)


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MaaiError("the HIP path needs tensors on a HIP device (got %s); there is no CPU fallback" % t.device)
        if t is not None and not t.is_contiguous():
            raise MaaiError("tensor must be contiguous")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------
# optional per-launch timing (HIP events on the launch stream) for bench.py's
# roofline: records (kernel, algorithmic flops, algorithmic bytes, start, end)
# ----------------------------------------------------------------------------
_PROF = None
DETAIL = [False]  # True: conv records carry their geometry in the name
FLOPS_SCALE = [1.0]  # set by the engine around the stem (executed K includes zero padding)


class profile(object):
    """with kernels.profile() as prof: ... ; prof.table() -> {wrapper name: dict(ms, flops, bytes, bytes_8d, launches)};
    prof.kernel_table() -> the same keyed by the DEVICE kernel's name as rocprofv3's kernel trace prints it (the library
    notes the kernel each compute call launched: maai_kernel_names / maai_last_kernel_name)."""

    def __enter__(self):
        global _PROF
        self.records = []
        _PROF = self.records
        self._names_were = lib().maai_kernel_names(1)
        return self

    def __exit__(self, *a):
        global _PROF
        _PROF = None
        lib().maai_kernel_names(self._names_were)
        torch.cuda.synchronize()

    def _table(self, key):
        out = {}
        for rec in self.records:
            name, flops, nbytes, b8d, e0, e1 = rec[key], rec[2], rec[3], rec[4], rec[5], rec[6]
            t = out.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, bytes_8d=0.0, launches=0))
            t["ms"] += e0.elapsed_time(e1)
            t["flops"] += flops
            t["bytes"] += nbytes
            t["bytes_8d"] += b8d
            t["launches"] += 1
        return out

    def table(self):
        """bytes = what the launch moves as built (every tensor its epilogue reads or writes); bytes_8d = SURVEY §8(d)'s
        algorithmic figure for a convolution: its input once + its output once."""
        return self._table(0)

    def kernel_table(self):
        return self._table(1)


def short_kernel_name(name):
    """the spelling scripts/profile_summary.py gives rocprofv3's names in profiles/*_kernel_stats.csv"""
    return name.replace("unsigned short", "bf16")


class _timed(object):
    __slots__ = ("name", "flops", "bytes", "b8d", "e0")

    def __init__(self, name, flops, nbytes, b8d=None):
        self.name, self.flops, self.bytes, self.b8d = name, flops, nbytes, (nbytes if b8d is None else b8d)

    def __enter__(self):
        if _PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if _PROF is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            kn = lib().maai_last_kernel_name()
            kn = short_kernel_name(kn.decode()) if kn else ""
            _PROF.append((self.name, kn or self.name, self.flops, self.bytes, self.b8d, self.e0, e1))


# ----------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------
def conv_out_hw(ih, iw, kh, kw, stride, pad_h, pad_w):
    return (ih + 2 * pad_h - kh) // stride + 1, (iw + 2 * pad_w - kw) // stride + 1


def make_desc(x, w, stride, pad_h, pad_w, grid_hw=None, out_hw=None, out_stride=1, out_off=(0, 0), accumulate=False, x2=None):
    n, ih, iw, cin = x.shape
    if x2 is not None:   # two-source input: the channels of x followed by those of x2
        if tuple(x2.shape[:3]) != (n, ih, iw) or x2.dtype != x.dtype:
            raise MaaiError("conv2d: the second input tensor must cover the same pixels in the same dtype")
        cin += x2.shape[3]
    cout, kh, kw, cin_w = w.shape
    if cin != cin_w:
        raise MaaiError("conv2d: Cin mismatch %d vs %d" % (cin, cin_w))
    oh, ow = conv_out_hw(ih, iw, kh, kw, stride, pad_h, pad_w) if grid_hw is None else grid_hw
    toh, tow = (oh, ow) if out_hw is None else out_hw
    return ConvDesc(n, ih, iw, cin, cout, kh, kw, stride, pad_h, pad_w, oh, ow, toh, tow, out_stride, out_off[0], out_off[1],
                    1 if accumulate else 0)


class Lazy(object):
    """An activation that exists only as the raw convolution output it is computed from: act(y*scale + shift), or
    the join act((y*scale + shift) + r(b*scale2 + shift2)) of two raw tensors (b added as is when scale2 is None).
    Consumers (conv2d / conv2d_wgrad with ``xf=``) form it on load (maai_conv_epilogue.xs ...)."""
    __slots__ = ("y", "scale", "shift", "relu", "b", "scale2", "shift2", "pre")

    def __init__(self, y, scale, shift, relu=True, b=None, scale2=None, shift2=None, pre=None):
        self.y, self.scale, self.shift, self.relu, self.b, self.scale2, self.shift2 = y, scale, shift, relu, b, scale2, shift2
        # pre = (x, w): ``y`` does not exist — it is the pointwise convolution conv(x, w) (x a tensor or a single-tensor
        # Lazy), whose statistics gave (scale, shift); a chained launch recomputes it (conv2d with join_out)
        self.pre = pre
        if pre is not None and (y is not None or b is None):
            raise MaaiError("Lazy: a recomputed tensor has no y and is one side of a join")

    @property
    def shape(self):
        if self.y is not None:
            return self.y.shape
        return tuple(self.b.shape)

    @property
    def dtype(self):
        return self.b.dtype if self.y is None else self.y.dtype


def _xf_epilogue(xf, join_out=None, join_bits=None):
    _gpu(xf.y, xf.scale, xf.shift, xf.b, xf.scale2, xf.shift2, join_out, join_bits)
    c = xf.shape[-1]
    for t in (xf.scale, xf.shift, xf.scale2, xf.shift2):
        if t is not None and (t.dtype != torch.float32 or t.numel() != c):
            raise MaaiError("normalise-on-load: per-channel coefficients must be fp32 [Cin]")
    if xf.b is not None and xf.y is not None and (xf.b.shape != xf.y.shape or xf.b.dtype != xf.y.dtype):
        raise MaaiError("normalise-on-load: the second tensor must match the first")
    if (xf.scale2 is None) != (xf.shift2 is None) or (xf.b is None and (xf.scale2 is not None or join_out is not None)):
        raise MaaiError("normalise-on-load: scale2/shift2/out belong to the two-tensor join")
    epi = ConvEpilogue(EPI_STORE, 0, None, None, None, None)
    epi.xs, epi.xt, epi.x_relu = xf.scale.data_ptr(), xf.shift.data_ptr(), 1 if xf.relu else 0
    if xf.b is not None:
        epi.xb = xf.b.data_ptr()
        if xf.scale2 is not None:
            epi.xs2, epi.xt2 = xf.scale2.data_ptr(), xf.shift2.data_ptr()
        if join_out is not None:
            epi.x_out = join_out.data_ptr()
        if join_bits is not None:
            epi.x_bits = join_bits.data_ptr()
    return epi


def conv2d(x, w, stride=1, pad_h=0, pad_w=0, stats=False, out=None, grid_hw=None, out_hw=None, out_stride=1,
           out_off=(0, 0), accumulate=False, relu_mask=None, join_out=False, join_bits=False):
    """y = conv(x, w) (NHWC / KHWC), optionally y *= (relu_mask > 0).  Returns y or (y, stats_partial[rows,2,Cout]).
    ``x`` may be a ``Lazy`` activation: it is then formed on load from its raw tensor(s) and never stored — except
    that a two-tensor join is handed back once when ``join_out`` (and its 1-bit ReLU mask when ``join_bits``):
    the return value then ends with (joined [, bits])."""
    if isinstance(x, Lazy):
        return _conv2d_lazy(x, w, stride, pad_h, pad_w, stats, join_out, join_bits)
    _gpu(x, w, out, relu_mask)
    if x.dtype != w.dtype:
        raise MaaiError("conv2d: x and w must share the storage dtype")
    d = make_desc(x, w, stride, pad_h, pad_w, grid_hw, out_hw, out_stride, out_off, accumulate)
    if out is None:
        if accumulate or out_stride != 1:
            raise MaaiError("conv2d: scatter/accumulate needs an explicit output tensor")
        out = torch.empty((d.N, d.OH, d.OW, d.Cout), dtype=x.dtype, device=x.device)
    part = None
    if stats:
        rows = lib().maai_conv2d_stats_rows(C.byref(d), _dt(x))
        part = torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device)
    m = d.N * d.OHg * d.OWg
    es = x.element_size()
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm M%d Cin%d Cout%d k%dx%d s%d os%d acc%d" % (m, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.out_stride, d.accumulate)
    if relu_mask is not None and (relu_mask.shape != out.shape or relu_mask.dtype != out.dtype):
        raise MaaiError("conv2d: relu_mask must have the output's shape and dtype")
    with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin * FLOPS_SCALE[0],
                es * (x.numel() + w.numel() + m * d.Cout * (1 + (1 if accumulate else 0) + (1 if relu_mask is not None else 0))),
                es * (x.numel() + m * d.Cout)):
        check(lib().maai_conv2d_igemm(C.byref(d), _p(x), _p(w), _p(out), _p(part), _p(relu_mask), _dt(x), _stream()),
              "maai_conv2d_igemm")
    return (out, part) if stats else out


def _conv2d_chained(xf, w, stats, join_bits, keep_y):
    """conv(join(conv(pre_x, pre_w) normalised by (scale, shift), b), w): one launch (csrc/conv_chain.hip).  Returns
    (out [, stats], joined [, bits] [, y]) — y, the recomputed raw tensor, only when ``keep_y``."""
    px, pw = xf.pre
    plazy = px if isinstance(px, Lazy) else None
    pt = plazy.y if plazy is not None else px
    if plazy is not None and (plazy.b is not None or plazy.pre is not None):
        raise MaaiError("conv2d: the producer of a chained launch takes a tensor or a single-tensor Lazy")
    b = xf.b
    _gpu(pt, pw, b, w)
    if not (pt.dtype == pw.dtype == b.dtype == w.dtype == torch.bfloat16):
        raise MaaiError("conv2d: chained launches are bf16")
    if tuple(pw.shape[1:3]) != (1, 1) or pw.shape[0] != b.shape[-1] or pw.shape[3] != pt.shape[-1] or pt.shape[:3] != b.shape[:3]:
        raise MaaiError("conv2d: the chained producer must be a pointwise convolution onto the join's channels")
    d = make_desc(b, w, 1, 0, 0)
    out = torch.empty((d.N, d.OH, d.OW, d.Cout), dtype=b.dtype, device=b.device)
    jo = torch.empty_like(b)
    jb = torch.empty((b.numel() // 8,), dtype=torch.uint8, device=b.device) if join_bits else None
    yk = torch.empty_like(b) if keep_y else None
    epi = _xf_epilogue(xf, jo, jb)
    epi.pre_x, epi.pre_w, epi.pre_cin = pt.data_ptr(), pw.data_ptr(), pt.shape[-1]
    if plazy is not None:
        _gpu(plazy.scale, plazy.shift)
        epi.pre_xs, epi.pre_xt, epi.pre_relu = plazy.scale.data_ptr(), plazy.shift.data_ptr(), 1 if plazy.relu else 0
    if yk is not None:
        epi.pre_y_out = yk.data_ptr()
    part = None
    if stats:
        rows = lib().maai_conv2d_stats_rows_fused(C.byref(d), C.byref(epi), BF16)
        part = torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=b.device)
    m = d.N * d.OHg * d.OWg
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[chain] M%d Cin%d Cout%d k1x1 s1 os1 acc0" % (m, d.Cin, d.Cout)
    nbytes = 2 * (pt.numel() + b.numel() * (2 + (1 if keep_y else 0)) + pw.numel() + w.numel() + m * d.Cout) + (0 if jb is None else jb.numel())
    flops = 2.0 * m * (d.Cout * d.Cin + d.Cin * pt.shape[-1])
    # SURVEY §8(d) bytes of what this launch replaces: this convolution's input once + output once (the recomputation
    # of the producer stands in for the read of its output)
    with _timed(nm, flops, nbytes, 2 * (b.numel() + m * d.Cout)):
        check(lib().maai_conv2d_igemm_fused(C.byref(d), None, _p(w), _p(out), _p(part), None, C.byref(epi), BF16, _stream()),
              "maai_conv2d_igemm_fused")
    ret = (out, part) if stats else (out,)
    ret = ret + (jo,)
    if join_bits:
        ret = ret + (jb,)
    if keep_y:
        ret = ret + (yk,)
    return ret


def _conv2d_lazy(xf, w, stride, pad_h, pad_w, stats, join_out, join_bits):
    if xf.pre is not None:
        raise MaaiError("conv2d: a recomputed input goes through conv2d_chained")
    x = xf.y
    _gpu(x, w)
    if x.dtype != w.dtype:
        raise MaaiError("conv2d: x and w must share the storage dtype")
    d = make_desc(x, w, stride, pad_h, pad_w)
    out = torch.empty((d.N, d.OH, d.OW, d.Cout), dtype=x.dtype, device=x.device)
    jo = torch.empty_like(x) if (join_out and xf.b is not None) else None
    jb = torch.empty((x.numel() // 8,), dtype=torch.uint8, device=x.device) if (jo is not None and join_bits and x.dtype == torch.bfloat16) else None
    epi = _xf_epilogue(xf, jo, jb)
    part = None
    if stats:
        rows = lib().maai_conv2d_stats_rows_fused(C.byref(d), C.byref(epi), _dt(x))
        part = torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device)
    m = d.N * d.OHg * d.OWg
    es = x.element_size()
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[xf%d] M%d Cin%d Cout%d k%dx%d s%d os1 acc0" % (2 if xf.b is not None else 1, m, d.Cin, d.Cout, d.KH, d.KW, d.stride)
    nbytes = es * (x.numel() * (1 + (1 if xf.b is not None else 0) + (1 if jo is not None else 0)) + w.numel() + m * d.Cout) + (0 if jb is None else jb.numel())
    with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin * FLOPS_SCALE[0], nbytes, es * (x.numel() + m * d.Cout)):
        check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(x), _p(w), _p(out), _p(part), None, C.byref(epi), _dt(x), _stream()),
              "maai_conv2d_igemm_fused")
    ret = (out, part) if stats else (out,)
    if join_out:
        ret = ret + (jo,)
        if join_bits:
            ret = ret + (jb,)
    return ret if len(ret) > 1 else ret[0]


def conv2d_stats_rows(x, w, stride=1, pad_h=0, pad_w=0, grid_hw=None, out_hw=None, out_stride=1, out_off=(0, 0), axf=False, x2=None):
    """Rows of the partial-sum slab one conv2d launch of this geometry writes (``axf``: transformed-operand launch)."""
    d = make_desc(x, w, stride, pad_h, pad_w, grid_hw, out_hw, out_stride, out_off, False, x2=x2)
    if axf:
        return (d.N * d.OHg * d.OWg + 127) // 128
    return int(lib().maai_conv2d_stats_rows(C.byref(d), _dt(x)))


def conv_module_family(conv, n, ih, iw, dtype):
    """the same for an nn.Conv2d module on an [n, ih, iw, Cin] input of ``dtype`` (no tensors needed)"""
    kh, kw = conv.kernel_size
    s, p = conv.stride[0], conv.padding[0]
    oh, ow = conv_out_hw(ih, iw, kh, kw, s, p, p)
    d = ConvDesc(int(n), int(ih), int(iw), conv.in_channels, conv.out_channels, kh, kw, s, p, p, oh, ow, oh, ow, 1, 0, 0, 0)
    return int(lib().maai_conv2d_kernel_family(C.byref(d), BF16 if dtype == torch.bfloat16 else F32))


def conv2d_kernel_family(x, w, stride=1, pad_h=0, pad_w=0):
    """0 ring / halo, 1 streaming, 2 ping-pong, 3 persistent 64-channel 3x3: the kernel a plain forward launch of this geometry gets (``x``: a tensor
    or a Lazy — only its shape and dtype matter)."""
    t = x.y if isinstance(x, Lazy) and x.y is not None else (x.b if isinstance(x, Lazy) else x)
    d = make_desc(t, w, stride, pad_h, pad_w)
    return int(lib().maai_conv2d_kernel_family(C.byref(d), _dt(t)))


def conv2d_store_reduce(x, w, stride, pad_h, pad_w, out, part, lower_y, mean, scale=None, shift=None, relu_mask=None,
                        grid_hw=None, out_hw=None, out_stride=1, out_off=(0, 0), accumulate=False, mask_bits=False, axf=None,
                        sum_increment=False, x2=None, bias=None, diag=None):
    """conv2d(..., out=out) for a data gradient whose epilogue also reduces the BatchNorm-backward partial sums of
    the stored values g: rows of ``part`` [rows,2,Cout] get sum(g) and sum(g*(lower_y - mean)).  The ReLU mask is
    ``relu_mask > 0`` or, without it, ``lower_y*scale + shift > 0`` (MAAI_EPI_DGRAD_REDUCE).
    ``axf = (y_raw, k1, k2, k3, dy_out)``: x is dz and the GEMM operand is k1*dz - k2 - k3*y_raw (BatchNorm-backward
    apply of the layer above), formed while staging; dy_out (or None) receives it.  Pointwise bf16 layers only.
    ``x2`` / ``bias``: the input is the channel concatenation [x | x2] of two tensors (w: [Cout,1,1,Cx+Cx2]) and ``bias`` [Cout]
    is added before rounding — the data gradient of a unit whose BatchNorm backward is folded (``fold_dgrad_weights(cat=True)``)."""
    _gpu(x, w, out, part, lower_y, mean, scale, shift, relu_mask, x2, bias, diag)
    if x.dtype != w.dtype or (lower_y is not None and (lower_y.dtype != out.dtype or lower_y.shape != out.shape)):
        raise MaaiError("conv2d_store_reduce: operand dtype / shape mismatch")
    if lower_y is None and relu_mask is None:
        # (the unit below folds its BatchNorm backward through its convolution: only sum(g) is wanted, the mask is its own)
        raise MaaiError("conv2d_store_reduce: without the lower layer's raw output the mask must be given (first sum only)")
    if mask_bits:
        if relu_mask is None or relu_mask.dtype != torch.uint8 or relu_mask.numel() * 8 != out.numel() or out.dtype != torch.bfloat16:
            raise MaaiError("conv2d_store_reduce: the 1-bit mask must be uint8 [numel/8] over a bf16 output")
    elif relu_mask is not None and (relu_mask.shape != out.shape or relu_mask.dtype != out.dtype):
        raise MaaiError("conv2d_store_reduce: relu_mask must have the output's shape and dtype")
    d = make_desc(x, w, stride, pad_h, pad_w, grid_hw, out_hw, out_stride, out_off, accumulate, x2=x2)
    epi = ConvEpilogue(EPI_DGRAD_REDUCE, 0, mean.data_ptr(), None if scale is None else scale.data_ptr(),
                       None if shift is None else shift.data_ptr(), None if lower_y is None else lower_y.data_ptr(), 1 if mask_bits else 0,
                       1 if (sum_increment and accumulate) else 0)
    if axf is not None:
        ya, k1, k2, k3, dyo = axf
        _gpu(ya, k1, k2, k3, dyo)
        if ya.shape != x.shape or ya.dtype != x.dtype or (dyo is not None and (dyo.shape != x.shape or dyo.dtype != x.dtype)):
            raise MaaiError("conv2d_store_reduce: the transformed operand's tensors must match x")
        epi.a2, epi.ak1, epi.ak2, epi.ak3 = ya.data_ptr(), k1.data_ptr(), k2.data_ptr(), k3.data_ptr()
        epi.a_out = None if dyo is None else dyo.data_ptr()
    if x2 is not None:
        epi.x2, epi.cin1 = x2.data_ptr(), x.shape[-1]
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != d.Cout:
            raise MaaiError("conv2d_store_reduce: bias is fp32 [Cout]")
        epi.bias = bias.data_ptr()
    if diag is not None:
        if lower_y is None or scale is None or shift is None or diag.dtype != torch.float32 or diag.numel() != d.Cout:
            raise MaaiError("conv2d_store_reduce: diag is fp32 [Cout] and needs the lower unit's raw output, scale and shift")
        epi.diag = diag.data_ptr()
    rows = int(lib().maai_conv2d_stats_rows_fused(C.byref(d), C.byref(epi), _dt(x)))
    if part.dtype != torch.float32 or not part.is_contiguous() or tuple(part.shape) != (rows, 2, d.Cout):
        raise MaaiError("conv2d_store_reduce: partial slab must be fp32 [%d, 2, %d]" % (rows, d.Cout))
    m = d.N * d.OHg * d.OWg
    es = x.element_size()
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[red] M%d Cin%d Cout%d k%dx%d s%d os%d acc%d" % (m, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.out_stride, d.accumulate)
    with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin * FLOPS_SCALE[0],
                es * (x.numel() * (1 if axf is None else (3 if axf[4] is not None else 2)) + (0 if x2 is None else x2.numel()) + w.numel()
                      + m * d.Cout * ((1 if lower_y is None else 2) + (1 if accumulate else 0) + (0 if (relu_mask is None or mask_bits) else 1)))
                + (m * d.Cout // 8 if mask_bits else 0), es * (x.numel() + m * d.Cout)):
        check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(x), _p(w), _p(out), _p(part), _p(relu_mask), C.byref(epi), _dt(x), _stream()),
              "maai_conv2d_igemm_fused")
    return out


def conv_bwd3(g, y3, y2, wd, k1, k2, k3, mean2, s2, t2, dx=None):
    """Backward of a 64 -> 256 bottleneck's conv3 + bn3 in one launch (csrc/conv_bwd3.hip): returns
    (dx [N,H,W,64], slab [rows,2,64] of bn2's backward partial sums, dw [256,1,1,64] fp32).  ``dx`` given: the data
    gradient is ADDED to it in place (before the mask; the sums are those of the stored result)."""
    _gpu(g, y3, y2, wd, k1, k2, k3, mean2, s2, t2, dx)
    if not (g.dtype == y3.dtype == y2.dtype == wd.dtype == torch.bfloat16) or g.shape != y3.shape or g.shape[-1] != 256 or \
            y2.shape[-1] != 64 or y2.shape[:-1] != g.shape[:-1] or tuple(wd.shape) != (64, 1, 1, 256):
        raise MaaiError("conv_bwd3: g, y3 [..,256], y2 [..,64] and wd [64,1,1,256] in bf16")
    m = g.numel() // 256
    rows = int(lib().maai_conv_bwd3_rows(m))
    acc = dx is not None
    if acc and (dx.shape != y2.shape or dx.dtype != y2.dtype or not dx.is_contiguous()):
        raise MaaiError("conv_bwd3: dx must match y2")
    if not acc:
        dx = torch.empty_like(y2)
    slab = torch.empty((rows, 2, 64), dtype=torch.float32, device=g.device)
    dw = torch.zeros((256, 1, 1, 64), dtype=torch.float32, device=g.device)
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[bwd3] M%d Cin256 Cout64 k1x1 s1 os1 acc0" % m
    with _timed(nm, 2.0 * m * 256 * 64 * 2, 2 * (2 * g.numel() + (3 if acc else 2) * y2.numel() + wd.numel()), 2 * (g.numel() + y2.numel())):
        check(lib().maai_conv_bwd3(_p(g), _p(y3), _p(y2), _p(wd), _p(k1), _p(k2), _p(k3), _p(mean2), _p(s2), _p(t2), _p(dx), _p(slab),
                                   _p(dw), m, 1 if acc else 0, _stream()), "maai_conv_bwd3")
    return dx, slab, dw


def _conv_fused(x, w, stride, pad_h, pad_w, mode, out, part, p0=None, p1=None, p2=None, t=None, relu=False, name="conv_igemm"):
    d = make_desc(x, w, stride, pad_h, pad_w)
    epi = ConvEpilogue(mode, 1 if relu else 0, None if p0 is None else p0.data_ptr(), None if p1 is None else p1.data_ptr(),
                       None if p2 is None else p2.data_ptr(), None if t is None else t.data_ptr())
    m = d.N * d.OHg * d.OWg
    es = x.element_size()
    nbytes = es * (x.numel() + w.numel() + (m * d.Cout if out is not None else 0) + (t.numel() if t is not None else 0))
    nm = name if not DETAIL[0] else "%s[epi%d] M%d Cin%d Cout%d k%dx%d" % (name, mode, m, d.Cin, d.Cout, d.KH, d.KW)
    with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin * FLOPS_SCALE[0], nbytes, es * (x.numel() + m * d.Cout)):
        check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(x), _p(w), _p(out), _p(part), None, C.byref(epi), _dt(x), _stream()),
              "maai_conv2d_igemm_fused")
    return d


def _stats_slab(x, w, stride, pad_h, pad_w, mode=EPI_STATS_ONLY):
    d = make_desc(x, w, stride, pad_h, pad_w)
    epi = ConvEpilogue(mode, 0, None, None, None, None)
    rows = lib().maai_conv2d_stats_rows_fused(C.byref(d), C.byref(epi), _dt(x))
    return torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=x.device), d


def conv2d_stats_only(x, w, stride=1, pad_h=0, pad_w=0):
    """BatchNorm partial statistics of conv(x, w) without storing the convolution (pass 1 of the fused unit; the
    producer side of a chained launch).  ``x`` may be a single-tensor ``Lazy`` where the streaming kernel takes the shape."""
    if isinstance(x, Lazy):
        if x.b is not None or x.pre is not None:
            raise MaaiError("conv2d_stats_only: a single-tensor Lazy")
        xt = x.y
        _gpu(xt, w)
        d = make_desc(xt, w, stride, pad_h, pad_w)
        epi = _xf_epilogue(x)
        epi.mode = EPI_STATS_ONLY
        rows = lib().maai_conv2d_stats_rows_fused(C.byref(d), C.byref(epi), _dt(xt))
        part = torch.empty((rows, 2, d.Cout), dtype=torch.float32, device=xt.device)
        m = d.N * d.OHg * d.OWg
        nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[stats-only xf1] M%d Cin%d Cout%d k1x1" % (m, d.Cin, d.Cout)
        with _timed(nm, 2.0 * m * d.Cout * d.Cin, xt.element_size() * (xt.numel() + w.numel()), 0.0):
            check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(xt), _p(w), None, _p(part), None, C.byref(epi), _dt(xt), _stream()),
                  "maai_conv2d_igemm_fused")
        return part
    _gpu(x, w)
    part, _ = _stats_slab(x, w, stride, pad_h, pad_w)
    _conv_fused(x, w, stride, pad_h, pad_w, EPI_STATS_ONLY, None, part)
    return part


def conv2d_chained(xf, w, stats=False, join_bits=False, keep_y=False):
    """See ``_conv2d_chained``; ``xf`` is a ``Lazy`` with ``pre`` set."""
    if not isinstance(xf, Lazy) or xf.pre is None:
        raise MaaiError("conv2d_chained: a Lazy carrying its producer")
    return _conv2d_chained(xf, w, stats, join_bits, keep_y)


def conv2d_bn_act(x, w, scale, shift, residual=None, relu=True, stride=1, pad_h=0, pad_w=0, want_bits=False):
    """out = act(conv(x, w)*scale + shift (+ residual)) in the conv epilogue (pass 2 of the fused unit; every unit of an
    inference forward with frozen statistics).  ``x`` may be a single-tensor ``Lazy`` where the streaming kernel takes the
    shape (``conv_bn_act_fast(..., lazy=True)``): formed on load AND normalised on store in one launch.
    ``want_bits`` (ReLU units on the streaming kernel, tensor input): also the 1-bit mask of out -> (out, bits)."""
    if isinstance(x, Lazy):
        if x.b is not None or x.pre is not None or want_bits:
            raise MaaiError("conv2d_bn_act: a single-tensor Lazy (and no mask output)")
        xt = x.y
        _gpu(xt, w, scale, shift, residual)
        d = make_desc(xt, w, stride, pad_h, pad_w)
        epi = _xf_epilogue(x)
        epi.mode, epi.relu = EPI_BN_ACT, 1 if relu else 0
        epi.p0, epi.p1, epi.t = scale.data_ptr(), shift.data_ptr(), (None if residual is None else residual.data_ptr())
        out = torch.empty((d.N, d.OH, d.OW, d.Cout), dtype=xt.dtype, device=xt.device)
        m = d.N * d.OHg * d.OWg
        es = xt.element_size()
        nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[xf1 epi2] M%d Cin%d Cout%d k%dx%d" % (m, d.Cin, d.Cout, d.KH, d.KW)
        with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin * FLOPS_SCALE[0],
                    es * (xt.numel() + w.numel() + m * d.Cout * (2 if residual is not None else 1)), es * (xt.numel() + m * d.Cout)):
            check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(xt), _p(w), _p(out), None, None, C.byref(epi), _dt(xt), _stream()),
                  "maai_conv2d_igemm_fused")
        return out
    _gpu(x, w, scale, shift, residual)
    d = make_desc(x, w, stride, pad_h, pad_w)
    out = torch.empty((d.N, d.OH, d.OW, d.Cout), dtype=x.dtype, device=x.device)
    if want_bits:
        if not relu or x.dtype != torch.bfloat16:
            raise MaaiError("conv2d_bn_act: the mask output belongs to a bf16 ReLU unit")
        bits = torch.empty((out.numel() // 8,), dtype=torch.uint8, device=x.device)
        epi = ConvEpilogue(EPI_BN_ACT, 1, scale.data_ptr(), shift.data_ptr(), None, None if residual is None else residual.data_ptr())
        epi.x_bits = bits.data_ptr()
        m = d.N * d.OHg * d.OWg
        nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[epi2 bits] M%d Cin%d Cout%d k%dx%d" % (m, d.Cin, d.Cout, d.KH, d.KW)
        with _timed(nm, 2.0 * m * d.Cout * d.KH * d.KW * d.Cin, 2 * (x.numel() + w.numel() + m * d.Cout * (2 if residual is not None else 1)) + bits.numel(),
                    2 * (x.numel() + m * d.Cout)):
            check(lib().maai_conv2d_igemm_fused(C.byref(d), _p(x), _p(w), _p(out), None, None, C.byref(epi), BF16, _stream()),
                  "maai_conv2d_igemm_fused")
        return out, bits
    _conv_fused(x, w, stride, pad_h, pad_w, EPI_BN_ACT, out, None, scale, shift, None, residual, relu)
    return out


def conv_bn_act_fast(conv, n, ih, iw, dtype, lazy=False):
    """Does the frozen-BatchNorm epilogue of this nn.Conv2d on an [n, ih, iw, Cin] input run on the kernel its plain launch
    would use (maai_conv2d_bn_act_fast)?  ``lazy``: with a normalise-on-load input."""
    kh, kw = conv.kernel_size
    s, p = conv.stride[0], conv.padding[0]
    pw = p if kw > 1 else 0
    oh, ow = conv_out_hw(ih, iw, kh, kw, s, p, pw)
    d = ConvDesc(int(n), int(ih), int(iw), conv.in_channels, conv.out_channels, kh, kw, s, p, pw, oh, ow, oh, ow, 1, 0, 0, 0)
    return bool(lib().maai_conv2d_bn_act_fast(C.byref(d), BF16 if dtype == torch.bfloat16 else F32, 1 if lazy else 0))


def conv2d_bwd_reduce(x, w, dz, mean, stride=1, pad_h=0, pad_w=0):
    """fp64 sums [2C] of dz and dz*(conv(x,w) - mean) with the convolution recomputed, never stored."""
    _gpu(x, w, dz, mean)
    part, _ = _stats_slab(x, w, stride, pad_h, pad_w)
    _conv_fused(x, w, stride, pad_h, pad_w, EPI_BWD_REDUCE, None, part, mean, None, None, dz)
    return reduce_partials(part)


def conv2d_bwd_apply(x, w, dz, k1, k2, k3, stride=1, pad_h=0, pad_w=0):
    """dy = k1*dz - k2 - k3*conv(x, w) with the convolution recomputed in the same kernel."""
    _gpu(x, w, dz, k1, k2, k3)
    dy = torch.empty_like(dz)
    _conv_fused(x, w, stride, pad_h, pad_w, EPI_BWD_APPLY, dy, None, k1, k2, k3, dz)
    return dy


_WGRAD_TUNE = {}
WGRAD_CANDIDATES = (768, 1536, 3072)
AUTOTUNE = [os.environ.get("MAAI_AUTOTUNE", "1") != "0"]
# MAAI_WGRAD_TUNE_FILE: the measured split-K budgets are read from / appended to this JSON file, so that a profiled
# run (rocprofv3) of a workload tuned before contains no tuning launches
_TUNE_FILE = os.environ.get("MAAI_WGRAD_TUNE_FILE", "")


def _tune_load():
    if _TUNE_FILE and os.path.exists(_TUNE_FILE):
        import json
        try:
            with open(_TUNE_FILE) as fh:
                for k, v in json.load(fh).items():
                    _WGRAD_TUNE[tuple(int(t) for t in k.split(","))] = int(v)
        except (ValueError, OSError):
            pass


def _tune_save():
    if _TUNE_FILE:
        import json
        tmp = "%s.%d.tmp" % (_TUNE_FILE, os.getpid())
        with open(tmp, "w") as fh:
            json.dump({",".join(str(t) for t in k): v for k, v in _WGRAD_TUNE.items()}, fh)
        os.replace(tmp, _TUNE_FILE)


_tune_load()


def _wgrad_target(d, x, dy, dtype_code):
    """Split-K workgroup budget for this shape: measured once (HIP events), then cached."""
    if dtype_code != BF16 or not AUTOTUNE[0] or torch.cuda.is_current_stream_capturing():
        return 0
    key = (d.N, d.IH, d.IW, d.Cin, d.Cout, d.KH, d.KW, d.stride)
    hit = _WGRAD_TUNE.get(key)
    if hit is not None:
        return hit
    tmp = torch.zeros((d.Cout, d.KH, d.KW, d.Cin), dtype=torch.float32, device=x.device)
    best, best_ms = 0, None
    for cand in WGRAD_CANDIDATES:
        ms = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().maai_conv2d_wgrad_tuned(C.byref(d), _p(x), _p(dy), _p(tmp), dtype_code, cand, _stream()), "maai_conv2d_wgrad_tuned")
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        m = min(ms[1:])
        if best_ms is None or m < best_ms:
            best, best_ms = cand, m
    _WGRAD_TUNE[key] = best
    _tune_save()
    return best


def conv2d_wgrad(x, dy, kh, kw, stride=1, pad_h=0, pad_w=0, target_blocks=None):
    """dw[Cout,KH,KW,Cin] fp32 for y = conv(x, w); dy dense [N,OH,OW,Cout].  ``x`` may be a single-tensor ``Lazy``
    activation (normalised on load).  ``target_blocks``: the split-K workgroup budget (default: measured per shape)."""
    xs = xt = None
    x_relu = 0
    if isinstance(x, Lazy):
        if x.b is not None:
            raise MaaiError("conv2d_wgrad: a two-tensor join must be materialised by its consumer convolution")
        _gpu(x.scale, x.shift)
        xs, xt, x_relu, x = x.scale, x.shift, 1 if x.relu else 0, x.y
    _gpu(x, dy)
    n, ih, iw, cin = x.shape
    _, oh, ow, cout = dy.shape
    d = ConvDesc(n, ih, iw, cin, cout, kh, kw, stride, pad_h, pad_w, oh, ow, oh, ow, 1, 0, 0, 0)
    target = _wgrad_target(d, x, dy, _dt(x)) if target_blocks is None else int(target_blocks)
    dw = torch.zeros((cout, kh, kw, cin), dtype=torch.float32, device=x.device)
    nm = "conv_wgrad" if not DETAIL[0] else "conv_wgrad M%d Cin%d Cout%d k%dx%d s%d" % (dy.numel() // cout, cin, cout, kh, kw, stride)
    with _timed(nm, 2.0 * dy.numel() * kh * kw * cin * FLOPS_SCALE[0], x.element_size() * (x.numel() + dy.numel()) + 4 * dw.numel()):
        check(lib().maai_conv2d_wgrad_xf(C.byref(d), _p(x), _p(dy), _p(dw), _dt(x), target, _p(xs), _p(xt), x_relu, _stream()),
              "maai_conv2d_wgrad_xf")
    return dw


# ----------------------------------------------------------------------------
# batch norm
# ----------------------------------------------------------------------------
def reduce_partials(partial):
    """[rows, ...] fp32 -> fp64 column sums (flattened trailing dims)."""
    _gpu(partial)
    rows = partial.shape[0]
    c2 = partial.numel() // rows
    sums = torch.empty(c2, dtype=torch.float64, device=partial.device)
    with _timed("reduce_partials", 0.0, 4.0 * partial.numel()):
        check(lib().maai_reduce_partials(_p(partial), rows, c2, _p(sums), _stream()), "maai_reduce_partials")
    return sums


def bn_finalize(sums, count, gamma, beta, running_mean, running_var, momentum, eps):
    c = sums.numel() // 2
    dev = sums.device
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
    check(lib().maai_bn_finalize(_p(sums), float(count), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                 float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(shift), c, _stream()),
          "maai_bn_finalize")
    return mean, invstd, scale, shift


def bn_pack_stats(sums, count):
    """fp64 sums [2C] over ``count`` samples -> fp32 [2C+1]: mean | M2 | count (the SyncBatchNorm message of one rank)"""
    _gpu(sums)
    c = sums.numel() // 2
    packed = torch.empty(2 * c + 1, dtype=torch.float32, device=sums.device)
    check(lib().maai_bn_pack_stats(_p(sums), float(count), _p(packed), c, _stream()), "maai_bn_pack_stats")
    return packed


def bn_finalize_gathered(gathered, gamma, beta, running_mean, running_var, momentum, eps):
    """``gathered``: [world, 2C+1] fp32 rows of ``bn_pack_stats`` (any row stride, unit column stride) -> what
    ``bn_finalize`` returns for the merged statistics (Chan's parallel variance in fp64) + the merged sample count as a
    device double (the ranks' batches may differ; it is never read back: ``bn_bwd_coeffs`` takes it as is)."""
    _gpu(gamma, beta, running_mean, running_var)
    if not gathered.is_cuda:
        raise MaaiError("the HIP path needs tensors on a HIP device (got %s); there is no CPU fallback" % gathered.device)
    if gathered.dim() != 2 or gathered.dtype != torch.float32 or gathered.stride(1) != 1:
        raise MaaiError("bn_finalize_gathered: expected a [world, 2C+1] fp32 matrix with contiguous rows")
    world, c = gathered.shape[0], (gathered.shape[1] - 1) // 2
    dev = gathered.device
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
    count = torch.empty(1, dtype=torch.float64, device=dev)
    check(lib().maai_bn_finalize_gathered(_p(gathered), world, gathered.stride(0), _p(gamma), _p(beta), _p(running_mean),
                                          _p(running_var), float(momentum), float(eps), _p(mean), _p(invstd), _p(scale),
                                          _p(shift), _p(count), c, _stream()), "maai_bn_finalize_gathered")
    return mean, invstd, scale, shift, count


BN_UPDATE_SLOT = [("running", "<u8"), ("stat_a", "<u8"), ("stat_b", "<u8"), ("n", "<i8"), ("momentum", "<f4"), ("reserved", "<i4")]   # maai_bn_update_slot


def bn_running_update_multi(items):
    """``items``: [(running tensor, stat_a, stat_b or None, momentum)]: running <- (1-m)*running + m*stat_a, then stat_b, for
    every entry in ONE launch (maai_bn_running_update_multi; every running tensor at most once)."""
    import numpy as np
    if not items:
        return
    tab = np.zeros(len(items), dtype=BN_UPDATE_SLOT)
    seen = set()
    for i, (r, a, b, mom) in enumerate(items):
        _gpu(r, a, b)
        if r.dtype != torch.float32 or a.dtype != torch.float32 or a.numel() != r.numel() or (b is not None and (b.dtype != torch.float32 or b.numel() != r.numel())):
            raise MaaiError("bn_running_update_multi: fp32 tensors of one size per entry")
        if r.data_ptr() in seen:
            raise MaaiError("bn_running_update_multi: a running buffer may appear once (two updates go into one entry)")
        seen.add(r.data_ptr())
        tab[i] = (r.data_ptr(), a.data_ptr(), 0 if b is None else b.data_ptr(), r.numel(), float(mom), 0)
    dev = items[0][0].device
    tdev = torch.from_numpy(tab.view(np.uint8)).to(dev)
    check(lib().maai_bn_running_update_multi(_p(tdev), len(items), _stream()), "maai_bn_running_update_multi")
    tdev.record_stream(torch.cuda.current_stream())


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    _gpu(running_mean, running_var)
    c = running_mean.numel()
    scale, shift = (torch.empty(c, dtype=torch.float32, device=running_mean.device) for _ in range(2))
    check(lib().maai_bn_eval_coeffs(_p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(scale), _p(shift),
                                    c, _stream()), "maai_bn_eval_coeffs")
    return scale, shift


def _bn_name(base, m, c, tag):
    """bench.py --detail: one table row per BatchNorm-pass shape"""
    return base if not DETAIL[0] else "%s M%d C%d %s" % (base, m, c, tag)


def bn_act_fwd(y, scale, shift, residual=None, relu=True, out=None, want_bits=False):
    """out = act(y*scale + shift (+ residual)); with ``want_bits`` (bf16) also the 1-bit ReLU mask of out
    (uint8 [numel/8]) -> (out, bits)."""
    _gpu(y, scale, shift, residual, out)
    c = y.shape[-1]
    m = y.numel() // c
    if out is None:
        out = torch.empty_like(y)
    bits = torch.empty((y.numel() // 8,), dtype=torch.uint8, device=y.device) if want_bits else None
    with _timed(_bn_name("bn_act_fwd", m, c, "res" if residual is not None else ""), 0.0, y.element_size() * y.numel() * (3 if residual is not None else 2) + (y.numel() // 8 if want_bits else 0)):
        check(lib().maai_bn_act_fwd_mask(_p(y), _p(scale), _p(shift), _p(residual), _p(out), _p(bits), m, c, 1 if relu else 0,
                                         _dt(y), _stream()), "maai_bn_act_fwd_mask")
    return (out, bits) if want_bits else out


def bn_act_fwd2(y, scale, shift, y2, scale2, shift2, relu=True, want_bits=False):
    """out = act((y*scale + shift) + round(y2*scale2 + shift2)) in one pass (two BatchNorm branches meeting)."""
    _gpu(y, scale, shift, y2, scale2, shift2)
    if y.shape != y2.shape or y.dtype != y2.dtype:
        raise MaaiError("bn_act_fwd2: the two branches must share shape and dtype")
    c = y.shape[-1]
    m = y.numel() // c
    out = torch.empty_like(y)
    bits = torch.empty((y.numel() // 8,), dtype=torch.uint8, device=y.device) if want_bits else None
    with _timed(_bn_name("bn_act_fwd", m, c, "two"), 0.0, y.element_size() * y.numel() * 3 + (y.numel() // 8 if want_bits else 0)):
        check(lib().maai_bn_act_fwd2(_p(y), _p(scale), _p(shift), _p(y2), _p(scale2), _p(shift2), _p(out), _p(bits), m, c,
                                     1 if relu else 0, _dt(y), _stream()), "maai_bn_act_fwd2")
    return (out, bits) if want_bits else out


def bn_act_bwd_apply2(dz, y, k, y2, kb):
    """(k1*dz - k2 - k3*y, k1b*dz - k2b - k3b*y2) with dz read once; k, kb = (k1, k2, k3) triples."""
    _gpu(dz, y, y2, *k, *kb)
    c = dz.shape[-1]
    m = dz.numel() // c
    dy, dy2 = torch.empty_like(dz), torch.empty_like(dz)
    with _timed(_bn_name("bn_bwd_apply", m, c, "two"), 0.0, dz.element_size() * dz.numel() * 5):
        check(lib().maai_bn_act_bwd_apply2(_p(dz), _p(y), _p(k[0]), _p(k[1]), _p(k[2]), _p(y2), _p(kb[0]), _p(kb[1]), _p(kb[2]),
                                           _p(dy), _p(dy2), m, c, _dt(dz), _stream()), "maai_bn_act_bwd_apply2")
    return dy, dy2


def bn_act_bwd_reduce(dout, out, y, mean, relu):
    """fp64 sums [2C]: sum dz, sum dz*(y-mean)."""
    _gpu(dout, out, y, mean)
    c = dout.shape[-1]
    m = dout.numel() // c
    rows = lib().maai_bn_bwd_rows(m, c, _dt(dout))
    part = torch.empty((rows, 2, c), dtype=torch.float32, device=dout.device)
    if y is None:
        part.zero_()
    nt = 1 + (1 if relu else 0) + (1 if y is not None else 0)
    with _timed(_bn_name("bn_bwd_reduce", m, c, "t%d" % nt), 0.0, dout.element_size() * dout.numel() * nt):
        check(lib().maai_bn_act_bwd_reduce(_p(dout), _p(out), _p(y), _p(mean), _p(part), m, c, 1 if relu else 0, _dt(dout),
                                           _stream()), "maai_bn_act_bwd_reduce")
    return reduce_partials(part)


def bn_bwd_coeffs(sums, count, gamma, mean, invstd):
    """``sums``: fp64 (local / reduced in fp64) or fp32 (the cross-rank exchange of SyncBatchNorm's backward);
    ``count``: a number, or the device double ``bn_finalize_gathered`` returned (SyncBatchNorm's merged count)"""
    c = mean.numel()
    dev = mean.device
    dgamma, dbeta, k1, k2, k3 = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(5))
    cdev = None
    if torch.is_tensor(count):
        if count.dtype != torch.float64 or count.numel() != 1 or not count.is_cuda:
            raise MaaiError("bn_bwd_coeffs: a device count is one fp64 element")
        cdev, count = count, 0.0
    if sums.dtype == torch.float32:
        check(lib().maai_bn_bwd_coeffs_f32(_p(sums), float(count), _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(k1),
                                           _p(k2), _p(k3), c, _p(cdev), _stream()), "maai_bn_bwd_coeffs_f32")
    else:
        check(lib().maai_bn_bwd_coeffs(_p(sums), float(count), _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(k1),
                                       _p(k2), _p(k3), c, _p(cdev), _stream()), "maai_bn_bwd_coeffs")
    return dgamma, dbeta, k1, k2, k3


def bn_act_bwd_apply(dout, out, y, k1, k2, k3, relu, want_dy=True, want_dz=False):
    _gpu(dout, out, y, k1, k2, k3)
    c = dout.shape[-1]
    m = dout.numel() // c
    dy = torch.empty_like(dout) if want_dy else None
    dz = torch.empty_like(dout) if want_dz else None
    nt = 1 + (1 if relu else 0) + (1 if k1 is not None else 0) + (1 if want_dy else 0) + (1 if want_dz else 0)
    with _timed(_bn_name("bn_bwd_apply", m, c, "t%d%s%s" % (nt, "+dy" if want_dy else "", "+dz" if want_dz else "")), 0.0, dout.element_size() * dout.numel() * nt):
        check(lib().maai_bn_act_bwd_apply(_p(dout), _p(out), _p(y), _p(k1), _p(k2), _p(k3), _p(dy), _p(dz), m, c,
                                          1 if relu else 0, _dt(dout), _stream()), "maai_bn_act_bwd_apply")
    return dy, dz


# ----------------------------------------------------------------------------
# BatchNorm-backward folded through an expanding pointwise convolution (csrc/fold.hip; engine._FOLD)
# ----------------------------------------------------------------------------
def fold_s2(wq, g1, s1, mean):
    """fp64 [Cout]: sum_k W[c,k] G1[c,k] - mean[c] s1[c]  (= sum g*(y - mean), y = x W^T never read)"""
    _gpu(wq, g1, s1, mean)
    cout, cin = g1.shape
    if wq.dtype != torch.bfloat16 or wq.numel() != cout * cin or g1.dtype != torch.float32 or s1.dtype != torch.float64:
        raise MaaiError("fold_s2: bf16 weights [Cout,Cin], fp32 G1, fp64 s1")
    s2 = torch.empty(cout, dtype=torch.float64, device=g1.device)
    check(lib().maai_fold_s2(_p(wq), _p(g1), _p(s1), _p(mean), _p(s2), cout, cin, _stream()), "maai_fold_s2")
    return s2


def fold_dw(wq, g1, gram, sx, k1, k2, k3):
    """fp32 [Cout,Cin]: k1*G1 - k2 (x) sx - k3*(W Gram)"""
    _gpu(wq, g1, gram, sx, k1, k2, k3)
    cout, cin = g1.shape
    if gram.dtype != torch.float32 or gram.numel() != cin * cin or sx.dtype != torch.float64 or sx.numel() < cin:
        raise MaaiError("fold_dw: fp32 Gram [Cin,Cin], fp64 sx [Cin]")
    dw = torch.empty((cout, cin), dtype=torch.float32, device=g1.device)
    check(lib().maai_fold_dw(_p(wq), _p(g1), _p(gram), _p(sx), _p(k1), _p(k2), _p(k3), _p(dw), cout, cin, _stream()), "maai_fold_dw")
    return dw


def fold_dgrad_weights(wq, k1, k2, k3, s1, sx, count, cat=False):
    """The folded data gradient's weights.  ``cat`` False: (wf [Cin,1,1,Cout] bf16 = k1*W in data-gradient form, tn [Cin,1,1,Cin]
    bf16 = -(W^T diag(k3) W), cn [Cin] fp32 = -(k2 W) minus the pixel mean of what rounding wf and tn adds to dx: ``s1`` = sum g
    [Cout], ``sx`` = colsum x [Cin] (fp64) over ``count`` pixels).  ``cat``: ONE matrix Wcat [Cin,1,1,Cout+Cin] = [wf | tn] with
    tn's DIAGONAL zeroed and returned in fp32 — (wcat, cn, dg): the two-source launch / ``conv_dfold`` add dg[k]*x[p][k] in
    their epilogue (that entry is the large one of its row and meets the mask's own channel: it is kept out of bf16)."""
    _gpu(wq, k1, k2, k3, s1, sx)
    cout, cin = wq.shape[0], wq.numel() // wq.shape[0]
    if s1.dtype != torch.float64 or sx.dtype != torch.float64 or s1.numel() < cout or sx.numel() < cin:
        raise MaaiError("fold_dgrad_weights: fp64 s1 [Cout], sx [Cin]")
    cn = torch.empty(cin, dtype=torch.float32, device=wq.device)
    scratch = torch.empty(cin, dtype=torch.float32, device=wq.device)
    if cat:
        wcat = torch.empty((cin, 1, 1, cout + cin), dtype=torch.bfloat16, device=wq.device)
        dg = torch.empty(cin, dtype=torch.float32, device=wq.device)
        tnp = C.c_void_p(wcat.data_ptr() + 2 * cout)
        check(lib().maai_fold_dgrad_w(_p(wq), _p(k1), _p(k2), _p(k3), _p(s1), _p(sx), float(count), _p(wcat), cout + cin, tnp, cout + cin,
                                      _p(cn), _p(dg), _p(scratch), cout, cin, _stream()), "maai_fold_dgrad_w")
        return wcat, cn, dg
    wf = torch.empty((cin, 1, 1, cout), dtype=torch.bfloat16, device=wq.device)
    tn = torch.empty((cin, 1, 1, cin), dtype=torch.bfloat16, device=wq.device)
    check(lib().maai_fold_dgrad_w(_p(wq), _p(k1), _p(k2), _p(k3), _p(s1), _p(sx), float(count), _p(wf), cout, _p(tn), cin, _p(cn),
                                  None, _p(scratch), cout, cin, _stream()), "maai_fold_dgrad_w")
    return wf, tn, cn


GRAM_CHANNELS = (64, 128, 256, 512)


def gram(x):
    """(Gram = x^T x [C,C] fp32, sx = colsum(x) [C] fp64) of an activation [.., C] bf16 — a tensor or a single-tensor ``Lazy``
    (formed on load).  csrc/gram.hip; accumulated with atomics: last-bit run-to-run differences, like every weight gradient."""
    xs = xt = None
    x_relu = 0
    if isinstance(x, Lazy):
        if x.b is not None or x.pre is not None:
            raise MaaiError("gram: a tensor or a single-tensor Lazy")
        _gpu(x.scale, x.shift)
        xs, xt, x_relu, x = x.scale, x.shift, 1 if x.relu else 0, x.y
    _gpu(x)
    c = x.shape[-1]
    if x.dtype != torch.bfloat16 or c not in GRAM_CHANNELS:
        raise MaaiError("gram: bf16 activations with 64, 128, 256 or 512 channels")
    m = x.numel() // c
    g = torch.zeros((c, c), dtype=torch.float32, device=x.device)
    sx = torch.zeros(c, dtype=torch.float64, device=x.device)
    with _timed("gram" if not DETAIL[0] else "gram M%d C%d" % (m, c), 2.0 * m * c * c, 2.0 * x.numel(), 0.0):
        check(lib().maai_gram(_p(x), m, c, _p(xs), _p(xt), x_relu, _p(g), _p(sx), None, _stream()), "maai_gram")
    return g, sx


def gram_deterministic(x):
    """(Gram = x^T x [C,C], sx = colsum(x) [C]) in fp64 from per-workgroup fp32 partials summed in a fixed order
    (maai_gram_partials + maai_reduce_partials): bit-reproducible, so it may feed FORWARD statistics (``fold_stats``)."""
    xs = xt = None
    x_relu = 0
    if isinstance(x, Lazy):
        if x.b is not None or x.pre is not None:
            raise MaaiError("gram: a tensor or a single-tensor Lazy")
        _gpu(x.scale, x.shift)
        xs, xt, x_relu, x = x.scale, x.shift, 1 if x.relu else 0, x.y
    _gpu(x)
    c = x.shape[-1]
    if x.dtype != torch.bfloat16 or c not in GRAM_CHANNELS:
        raise MaaiError("gram: bf16 activations with 64, 128, 256 or 512 channels")
    m = x.numel() // c
    rows = int(lib().maai_gram_partial_rows(m, c))
    pg = torch.empty((rows, c * c), dtype=torch.float32, device=x.device)
    ps = torch.empty((rows, c), dtype=torch.float32, device=x.device)
    with _timed("gram" if not DETAIL[0] else "gram[det] M%d C%d" % (m, c), 2.0 * m * c * c, 2.0 * x.numel(), 0.0):
        check(lib().maai_gram_partials(_p(x), m, c, _p(xs), _p(xt), x_relu, _p(pg), _p(ps), _stream()), "maai_gram_partials")
    return reduce_partials(pg).reshape(c, c), reduce_partials(ps)


def fold_stats(wq, gram64, sx64):
    """fp64 [2*Cout] = sum y | sum y^2 of y = x W^T from Gram(x) and colsum(x) (``gram_deterministic``): what
    ``reduce_partials`` of a statistics-only launch returns, without the launch."""
    _gpu(wq, gram64, sx64)
    cout, cin = wq.shape[0], wq.numel() // wq.shape[0]
    if wq.dtype != torch.bfloat16 or gram64.dtype != torch.float64 or sx64.dtype != torch.float64 or gram64.numel() != cin * cin or sx64.numel() != cin:
        raise MaaiError("fold_stats: bf16 weights [Cout,Cin], fp64 Gram [Cin,Cin] and column sums [Cin]")
    sums = torch.empty(2 * cout, dtype=torch.float64, device=wq.device)
    check(lib().maai_fold_stats(_p(wq), _p(gram64), _p(sx64), _p(sums), cout, cin, _stream()), "maai_fold_stats")
    return sums


def conv_dfold(g, y2, wcat, cn, mean2, s2, t2, dx=None, dg=None):
    """The folded 64 -> 256 unit's data gradient (csrc/conv_dfold.hip): dx = ([g | relu(bn2(y2))] Wcat^T + cn + dg*a2) * [a2 > 0] with
    the unit below's BatchNorm-backward partial sums -> (dx [.., 64], slab [rows, 2, 64]).  ``dx`` given: += in place."""
    _gpu(g, y2, wcat, cn, mean2, s2, t2, dx, dg)
    if not (g.dtype == y2.dtype == wcat.dtype == torch.bfloat16) or g.shape[-1] != 256 or y2.shape[-1] != 64 or \
            y2.shape[:-1] != g.shape[:-1] or tuple(wcat.shape) != (64, 1, 1, 320):
        raise MaaiError("conv_dfold: g [..,256], y2 [..,64] and Wcat [64,1,1,320] in bf16")
    m = g.numel() // 256
    acc = dx is not None
    if acc and (dx.shape != y2.shape or dx.dtype != y2.dtype or not dx.is_contiguous()):
        raise MaaiError("conv_dfold: dx must match y2")
    if not acc:
        dx = torch.empty_like(y2)
    rows = int(lib().maai_conv_dfold_rows(m))
    slab = torch.empty((rows, 2, 64), dtype=torch.float32, device=g.device)
    nm = "conv_igemm" if not DETAIL[0] else "conv_igemm[dfold] M%d Cin320 Cout64 k1x1 s1 os1 acc%d" % (m, 1 if acc else 0)
    with _timed(nm, 2.0 * m * 320 * 64, 2 * (g.numel() + (3 if acc else 2) * y2.numel() + wcat.numel()), 2 * (g.numel() + y2.numel())):
        check(lib().maai_conv_dfold(_p(g), _p(y2), _p(wcat), _p(cn), _p(dg), _p(mean2), _p(s2), _p(t2), _p(dx), _p(slab), m, 1 if acc else 0,
                                    _stream()), "maai_conv_dfold")
    return dx, slab


# ----------------------------------------------------------------------------
# layout / pooling / casts
# ----------------------------------------------------------------------------
def pack_views_u8(views, cpad, dtype):
    for v in views:
        _gpu(v)
        if v.dtype != torch.uint8:
            raise MaaiError("pack_views_u8: views must be uint8 [B,H,W,3]")
    b, h, w, c3 = views[0].shape
    if c3 != 3:
        raise MaaiError("pack_views_u8: last dim must be 3")
    out = torch.empty((b, h, w, cpad), dtype=dtype, device=views[0].device)
    arr = (C.c_void_p * len(views))(*[v.data_ptr() for v in views])
    check(lib().maai_pack_views_u8(arr, len(views), b, h, w, cpad, _p(out), _dt(out), _stream()), "maai_pack_views_u8")
    return out


def stem_unroll(x, dtype):
    """x: NCHW fp32 [B,3,H,W] or u8 HWC [B,H,W,3] -> [B,H,W,32] (kw-unrolled stem operand)."""
    _gpu(x)
    if x.dtype == torch.uint8:
        b, h, w, _ = x.shape
        out = torch.empty((b, h, w, 32), dtype=dtype, device=x.device)
        check(lib().maai_stem_unroll_u8(_p(x), b, h, w, _p(out), _dt(out), _stream()), "maai_stem_unroll_u8")
    else:
        b, c, h, w = x.shape
        if c != 3 or x.dtype != torch.float32:
            raise MaaiError("stem_unroll: expects fp32 NCHW with 3 channels")
        out = torch.empty((b, h, w, 32), dtype=dtype, device=x.device)
        check(lib().maai_stem_unroll_nchw_f32(_p(x), b, h, w, _p(out), _dt(out), _stream()), "maai_stem_unroll_nchw_f32")
    return out


def nchw_to_nhwc(x, cpad, dtype):
    _gpu(x)
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, cpad), dtype=dtype, device=x.device)
    check(lib().maai_nchw_f32_to_nhwc(_p(x), b, c, h, w, cpad, _p(out), _dt(out), _stream()), "maai_nchw_f32_to_nhwc")
    return out


def nhwc_to_nchw(x, c):
    _gpu(x)
    b, h, w, cpad = x.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    check(lib().maai_nhwc_to_nchw_f32(_p(x), b, c, h, w, cpad, _p(out), _dt(x), _stream()), "maai_nhwc_to_nchw_f32")
    return out


def avgpool_fwd(x, ph, pw):
    _gpu(x)
    b, h, w, c = x.shape
    out = torch.empty((b, ph, pw, c), dtype=x.dtype, device=x.device)
    check(lib().maai_avgpool_fwd(_p(x), b, h, w, c, ph, pw, _p(out), _dt(x), _stream()), "maai_avgpool_fwd")
    return out


def avgpool_bwd(dout, h, w):
    _gpu(dout)
    b, ph, pw, c = dout.shape
    dx = torch.empty((b, h, w, c), dtype=dout.dtype, device=dout.device)
    check(lib().maai_avgpool_bwd(_p(dout), b, h, w, c, ph, pw, _p(dx), _dt(dout), _stream()), "maai_avgpool_bwd")
    return dx


def cast_from_f32(src, dtype):
    _gpu(src)
    if dtype == torch.float32:
        return src
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib().maai_cast_from_f32(_p(src), _p(out), src.numel(), _dt(out), _stream()), "maai_cast_from_f32")
    return out


def cast_to_f32(src):
    _gpu(src)
    if src.dtype == torch.float32:
        return src
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib().maai_cast_to_f32(_p(src), _p(out), src.numel(), _dt(src), _stream()), "maai_cast_to_f32")
    return out


# ----------------------------------------------------------------------------
# NT-Xent
# ----------------------------------------------------------------------------
def ntxent_normalize(h, normalize=True):
    _gpu(h)
    b, d = h.shape
    z = torch.empty_like(h)
    inv = torch.empty(b, dtype=torch.float32, device=h.device)
    check(lib().maai_ntxent_normalize(_p(h), _p(z), _p(inv), b, d, 1 if normalize else 0, _stream()), "maai_ntxent_normalize")
    return z, inv


def ntxent_fwd(z1, z2, Z1, Z2, temperature, row_offset):
    _gpu(z1, z2, Z1, Z2)
    b, d = z1.shape
    n = Z1.shape[0]
    loss = torch.empty((), dtype=torch.float32, device=z1.device)
    logits = torch.empty((b, n), dtype=torch.float32, device=z1.device)
    lse = torch.empty((4, b), dtype=torch.float32, device=z1.device)
    check(lib().maai_ntxent_fwd(_p(z1), _p(z2), _p(Z1), _p(Z2), _p(loss), _p(logits), _p(lse), b, n, d, float(temperature),
                                int(row_offset), _stream()), "maai_ntxent_fwd")
    return loss, logits, lse


def ntxent_bwd(z1, z2, Z1, Z2, lse, gloss, temperature, row_offset, local_in_gathered, want_dz1):
    _gpu(z1, z2, Z1, Z2, lse, gloss)
    b, d = z1.shape
    n = Z1.shape[0]
    dz2 = torch.empty_like(z2)
    dz1 = torch.empty_like(z1) if want_dz1 else None
    check(lib().maai_ntxent_bwd(_p(z1), _p(z2), _p(Z1), _p(Z2), _p(lse), _p(gloss), _p(dz1), _p(dz2), b, n, d,
                                float(temperature), int(row_offset), 1 if local_in_gathered else 0, _stream()),
          "maai_ntxent_bwd")
    return dz1, dz2


def ntxent_normalize_bwd(z, dz, inv, normalize=True):
    _gpu(z, dz, inv)
    b, d = z.shape
    dh = torch.empty_like(z)
    check(lib().maai_ntxent_normalize_bwd(_p(z), _p(dz), _p(inv), _p(dh), b, d, 1 if normalize else 0, _stream()),
          "maai_ntxent_normalize_bwd")
    return dh


# ----------------------------------------------------------------------------
# optimiser / augmentation
# ----------------------------------------------------------------------------
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _gpu(p, g, m, v)
    check(lib().maai_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                               int(step), float(grad_scale), _stream()), "maai_adam_step")


WEIGHT_FORM_DTYPE = [("w", "<u8"), ("out", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("KH", "<i4"), ("KW", "<i4"), ("mode", "<i4"),
                     ("nkh", "<i4"), ("nkw", "<i4"), ("cin_pad", "<i4"), ("khs", "<i4", (8,)), ("kws", "<i4", (8,)), ("dtype", "<i4"),
                     ("reserved", "<i4")]   # maai_weight_form, 120 bytes


class WeightForms(object):
    """Registry of kernel-layout weight copies refreshed by ONE launch (maai_weight_forms).  ``add`` registers a
    form and returns its output tensor; ``run`` converts the listed forms (all by default)."""
    CHUNK = 1024

    def __init__(self):
        self.entries = []      # (param, out, mode, khs, kws, cin_pad)
        self._tables = None    # (forms_dev, block_form_dev, block_first_dev, nblocks, ptrs) for the whole registry

    def add(self, param, dtype, mode, khs, kws, cin_pad=None):
        import numpy as np
        if param.dim() != 4 or param.dtype != torch.float32 or not param.is_contiguous():
            raise MaaiError("weight form: expected a contiguous fp32 [Cout,Cin,KH,KW] parameter")
        if len(khs) > 8 or len(kws) > 8 or not khs or not kws:
            raise MaaiError("weight form: 1..8 taps per axis")
        _gpu(param)
        co, ci = param.shape[0], param.shape[1]
        cp = ci if cin_pad is None else cin_pad
        shape = (co, len(khs), len(kws), cp) if mode == 0 else (ci, len(khs), len(kws), co)
        out = torch.empty(shape, dtype=dtype, device=param.device)
        self.entries.append((param, out, mode, tuple(khs), tuple(kws), cp))
        self._tables = None
        return len(self.entries) - 1, out

    def _build(self, idxs):
        import numpy as np
        forms = np.zeros(len(idxs), dtype=np.dtype(WEIGHT_FORM_DTYPE))
        slot, first = [], []
        for j, i in enumerate(idxs):
            param, out, mode, khs, kws, cp = self.entries[i]
            f = forms[j]
            f["w"], f["out"] = param.data_ptr(), out.data_ptr()
            f["Cout"], f["Cin"], f["KH"], f["KW"] = param.shape
            f["mode"], f["nkh"], f["nkw"], f["cin_pad"] = mode, len(khs), len(kws), cp
            f["khs"][:len(khs)] = khs
            f["kws"][:len(kws)] = kws
            f["dtype"] = BF16 if out.dtype == torch.bfloat16 else F32
            nb = (out.numel() + self.CHUNK - 1) // self.CHUNK
            slot.append(np.full(nb, j, dtype=np.int32))
            first.append(np.arange(nb, dtype=np.int64) * self.CHUNK)
        dev = self.entries[idxs[0]][0].device
        fd = torch.from_numpy(forms.view(np.uint8).reshape(-1)).to(dev)
        bs = torch.from_numpy(np.concatenate(slot)).to(dev)
        bf = torch.from_numpy(np.concatenate(first)).to(dev)
        return fd, bs, bf, int(bs.numel()), tuple(self.entries[i][0].data_ptr() for i in idxs)

    def run(self, idxs=None):
        if not self.entries:
            return
        if idxs is None:
            ptrs = tuple(e[0].data_ptr() for e in self.entries)
            if self._tables is None or self._tables[4] != ptrs:
                self._tables = self._build(list(range(len(self.entries))))
            t = self._tables
        else:
            t = self._build(list(idxs))
        check(lib().maai_weight_forms(_p(t[0]), _p(t[1]), _p(t[2]), t[3], _stream()), "maai_weight_forms")
        if idxs is not None:
            # the temporary tables must outlive the launch; the caching allocator keeps them stream-ordered
            t[0].record_stream(torch.cuda.current_stream())


class AdamMulti(object):
    """One-launch Adam over a fixed list of fp32 tensors (maai_adam_step_multi).  The block map depends on the sizes
    only and is built once; the slot table is re-uploaded per step because gradient tensors are new every step."""
    CHUNK = 2048

    def __init__(self, params, ms, vs):
        import numpy as np
        self.params, self.ms, self.vs = list(params), list(ms), list(vs)
        _gpu(*self.params, *self.ms, *self.vs)
        dev = self.params[0].device
        slot, first = [], []
        for i, p in enumerate(self.params):
            n = p.numel()
            nb = (n + self.CHUNK - 1) // self.CHUNK
            slot.append(np.full(nb, i, dtype=np.int32))
            first.append(np.arange(nb, dtype=np.int64) * self.CHUNK)
        self.block_slot = torch.from_numpy(np.concatenate(slot)).to(dev)
        self.block_first = torch.from_numpy(np.concatenate(first)).to(dev)
        self.nblocks = int(self.block_slot.numel())
        self.table = np.zeros((len(self.params), 5), dtype=np.int64)   # maai_adam_slot: p, g, m, v, n
        for i, (p, m, v) in enumerate(zip(self.params, self.ms, self.vs)):
            self.table[i, 0], self.table[i, 2], self.table[i, 3], self.table[i, 4] = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        self.table_dev = torch.empty((len(self.params), 5), dtype=torch.int64, device=dev)

    def step(self, grads, lr, beta1, beta2, eps, step, grad_scale=1.0):
        _gpu(*grads)
        for i, g in enumerate(grads):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.table[i, 4]:
                raise MaaiError("adam_step_multi: gradients must be contiguous fp32 tensors of the parameter's size")
            self.table[i, 1] = g.data_ptr()
        self.table_dev.copy_(torch.from_numpy(self.table), non_blocking=False)
        check(lib().maai_adam_step_multi(_p(self.table_dev), _p(self.block_slot), _p(self.block_first), self.nblocks, float(lr),
                                         float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream()),
              "maai_adam_step_multi")


class SgdMulti(object):
    """One-launch SGD over a fixed list of fp32 tensors (maai_sgd_step_multi); block map as in AdamMulti."""
    CHUNK = 2048

    def __init__(self, params, moms):
        import numpy as np
        self.params, self.moms = list(params), list(moms)
        _gpu(*self.params, *self.moms)
        dev = self.params[0].device
        slot, first = [], []
        for i, p in enumerate(self.params):
            nb = (p.numel() + self.CHUNK - 1) // self.CHUNK
            slot.append(np.full(nb, i, dtype=np.int32))
            first.append(np.arange(nb, dtype=np.int64) * self.CHUNK)
        self.block_slot = torch.from_numpy(np.concatenate(slot)).to(dev)
        self.block_first = torch.from_numpy(np.concatenate(first)).to(dev)
        self.nblocks = int(self.block_slot.numel())
        self.table = np.zeros((len(self.params), 5), dtype=np.int64)   # maai_adam_slot: p, g, m, v (unused), n
        for i, (p, m) in enumerate(zip(self.params, self.moms)):
            self.table[i, 0], self.table[i, 2], self.table[i, 4] = p.data_ptr(), m.data_ptr(), p.numel()
        self.table_dev = torch.empty((len(self.params), 5), dtype=torch.int64, device=dev)

    def step(self, grads, lr, momentum, weight_decay, first):
        _gpu(*grads)
        for i, g in enumerate(grads):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.table[i, 4]:
                raise MaaiError("sgd_step_multi: gradients must be contiguous fp32 tensors of the parameter's size")
            self.table[i, 1] = g.data_ptr()
        self.table_dev.copy_(torch.from_numpy(self.table), non_blocking=False)
        check(lib().maai_sgd_step_multi(_p(self.table_dev), _p(self.block_slot), _p(self.block_first), self.nblocks, float(lr),
                                        float(momentum), float(weight_decay), 1 if first else 0, _stream()), "maai_sgd_step_multi")


class LarcMulti(object):
    """||w||, ||g|| of every tensor in one launch (maai_multi_sqnorm) and the LARC rescaling of all gradients in a
    second (maai_larc_scale).  Block map built once from the sizes; the slot table is re-uploaded per step (gradient
    tensors are new every step)."""
    CHUNK = 2048

    def __init__(self, params):
        import numpy as np
        self.params = list(params)
        _gpu(*self.params)
        dev = self.params[0].device
        slot, first = [], []
        for i, p in enumerate(self.params):
            nb = (p.numel() + self.CHUNK - 1) // self.CHUNK
            slot.append(np.full(nb, i, dtype=np.int32))
            first.append(np.arange(nb, dtype=np.int64) * self.CHUNK)
        self.block_slot = torch.from_numpy(np.concatenate(slot)).to(dev)
        self.block_first = torch.from_numpy(np.concatenate(first)).to(dev)
        self.nblocks = int(self.block_slot.numel())
        self.table = np.zeros((len(self.params), 5), dtype=np.int64)
        for i, p in enumerate(self.params):
            self.table[i, 0], self.table[i, 4] = p.data_ptr(), p.numel()
        self.table_dev = torch.empty((len(self.params), 5), dtype=torch.int64, device=dev)
        self.norms = torch.empty((len(self.params), 2), dtype=torch.float64, device=dev)

    def _upload(self, grads):
        _gpu(*grads)
        for i, g in enumerate(grads):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.table[i, 4]:
                raise MaaiError("LARC: gradients must be contiguous fp32 tensors of the parameter's size")
            self.table[i, 1] = g.data_ptr()
        self.table_dev.copy_(torch.from_numpy(self.table), non_blocking=False)

    def sqnorms(self, grads):
        """[n,2] fp64: ||p||^2, ||g||^2 per tensor"""
        self._upload(grads)
        check(lib().maai_multi_sqnorm(_p(self.table_dev), _p(self.block_slot), _p(self.block_first), self.nblocks, len(self.params),
                                      _p(self.norms), _stream()), "maai_multi_sqnorm")
        return self.norms

    def scale(self, trust, lr, weight_decay, eps, clip):
        """in place on the gradients given to the last sqnorms() call"""
        check(lib().maai_larc_scale(_p(self.table_dev), _p(self.block_slot), _p(self.block_first), self.nblocks, _p(self.norms),
                                    float(trust), float(lr), float(weight_decay), float(eps), 1 if clip else 0, _stream()),
              "maai_larc_scale")


def softmax_ce_fwd(logits, labels, ncls):
    """mean cross-entropy over the first ``ncls`` columns of logits [B, ld] (fp32), labels int64 [B] -> (loss 0-d, lse [B])"""
    _gpu(logits, labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.int64 or logits.dim() != 2 or labels.shape != (logits.shape[0],):
        raise MaaiError("softmax_ce: logits must be fp32 [B,ld] and labels int64 [B]")
    b, ld = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    lse = torch.empty(b, dtype=torch.float32, device=logits.device)
    check(lib().maai_softmax_ce_fwd(_p(logits), _p(labels), _p(loss), _p(lse), b, int(ncls), ld, _stream()), "maai_softmax_ce_fwd")
    return loss, lse


def softmax_ce_bwd(logits, labels, lse, gloss, ncls):
    _gpu(logits, labels, lse, gloss)
    b, ld = logits.shape
    d = torch.empty_like(logits)
    check(lib().maai_softmax_ce_bwd(_p(logits), _p(labels), _p(lse), _p(gloss), _p(d), b, int(ncls), ld, _stream()), "maai_softmax_ce_bwd")
    return d


def sgd_step(p, g, mom, lr, momentum, weight_decay, first_step):
    _gpu(p, g, mom)
    check(lib().maai_sgd_step(_p(p), _p(g), _p(mom), p.numel(), float(lr), float(momentum), float(weight_decay),
                              1 if first_step else 0, _stream()), "maai_sgd_step")


def augment_view_u8(images, params, oh, ow):
    _gpu(images, params)
    if params.dtype != torch.float32 or params.dim() != 2 or params.shape[1] != 16 or params.shape[0] != images.shape[0]:
        raise MaaiError("augment_view_u8: params must be float32 [B,16]")
    b, h, w, _ = images.shape
    out = torch.empty((b, oh, ow, 3), dtype=torch.uint8, device=images.device)
    check(lib().maai_augment_view_u8(_p(images), _p(params), b, h, w, oh, ow, _p(out), _stream()), "maai_augment_view_u8")
    return out


def augment_params(b, h, w, seed, view, device, min_area=0.1, brightness=1.0, contrast=1.0, saturation=0.5, hue=90.0):
    """[b,16] f32 crop / flip / colour-twist parameters (defaults: the driver's, Contrastive_Learning.py:164-171)."""
    params = torch.empty((b, 16), dtype=torch.float32, device=device)
    _gpu(params)
    check(lib().maai_augment_params(_p(params), b, h, w, int(seed), int(view), float(min_area), float(brightness),
                                    float(contrast), float(saturation), float(hue), _stream()), "maai_augment_params")
    return params


def foveate_views_u8(images, params, out_size=30):
    """images [B,H,W,3] u8, params [B,32] f32 -> 4 views [B,out_size,out_size,3] u8 (crops 400/240/100/30)."""
    _gpu(images, params)
    if images.dtype != torch.uint8 or params.dtype != torch.float32 or params.shape[1] != 32:
        raise MaaiError("foveate_views_u8: images must be uint8 [B,H,W,3], params float32 [B,32]")
    b, h, w, _ = images.shape
    out = torch.empty((4, b, out_size, out_size, 3), dtype=torch.uint8, device=images.device)
    check(lib().maai_foveate_views_u8(_p(images), _p(params), b, h, w, out_size, _p(out), _stream()), "maai_foveate_views_u8")
    return [out[0], out[1], out[2], out[3]]
