"""The 8-wave ping-pong convolution kernel (csrc/conv_pp.hip: 256 x 256 tiles, 64-deep K-tiles, the two halves of the
workgroup one barrier apart) against the 4-wave ring kernel (csrc/conv_igemm.h), which test_gpu_kernels.py pins to the
oracle / fp64 torch: same K order, same MFMA sequence per output element, so outputs must be BIT-IDENTICAL — 3x3 and 1x1
layers, stride 1 and 2, two to 72 K-tiles (even and odd counts: the two-tile loop body and its tail), rows past M, one
and several column tiles — and the BatchNorm statistics slab (one row per 256 pixels) must agree to fp32 summation
order.  The data-gradient epilogue (mask from the unit below + BatchNorm-backward sums, optional accumulate and 1-bit
mask) likewise.  Reference semantics: resnet.py:20-28 (conv3x3 / conv1x1) and their gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from maai_hip import kernels
    return kernels


class env(object):
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# N, H, W, Cin, Cout, k, stride
CASES = [
    (2, 14, 14, 256, 256, 3, 1),     # 36 K-tiles, M = 392: a full and a ragged row tile
    (1, 9, 11, 64, 256, 3, 1),       # M = 99 < one tile, 9 K-tiles (odd: the three-tile tail)
    (3, 15, 15, 128, 512, 3, 1),     # two column tiles, 18 K-tiles
    (2, 16, 16, 256, 256, 1, 1),     # pointwise, 4 K-tiles
    (2, 12, 12, 128, 256, 1, 1),     # the minimum: 2 K-tiles (no steady-state trip at all)
    (2, 16, 16, 192, 256, 1, 1),     # 3 K-tiles
    (1, 28, 28, 512, 512, 3, 1),     # 72 K-tiles
    (2, 17, 17, 128, 256, 3, 2),     # stride 2, odd extent
    (4, 8, 8, 1024, 256, 1, 1),      # channel-reducing pointwise, M = 256 exactly
    (1, 20, 20, 320, 768, 3, 1),     # 5 K-tiles per tap (45 in all), three column tiles
    # more tiles than CUs: the persistent form (conv_ppp.hip) walks several tiles per workgroup — epilogue behind the next
    # tile's prologue, stores counted in its first waits
    (8, 96, 96, 256, 256, 1, 1),     # 288 tiles, 4 K-tiles (the minimum it takes)
    (7, 97, 99, 64, 512, 3, 1),      # 263 row tiles x 2 column tiles, the last row tile ragged, 9 K-tiles
    # 128-channel outputs: the 4 x 2 wave grid (512 x 128 tiles, all 160 KB of LDS)
    (2, 14, 14, 128, 128, 3, 1),     # M = 392 < one tile
    (3, 23, 23, 128, 128, 3, 1),     # M = 1587: three full tiles and a ragged one
    (1, 16, 16, 256, 128, 1, 1),     # pointwise
    (2, 17, 17, 128, 384, 3, 2),     # Cout = 384: three column tiles of 128, stride 2
]


def _rows(cout):
    return 256 if cout % 256 == 0 else 512


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_pingpong_forward_matches_ring_kernel(K, case):
    n, h, w_, cin, cout, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 10007)
    x = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).cuda().bfloat16()
    pad = k // 2
    with env(MAAI_CONV_PP="0"):
        y0, st0 = K.conv2d(x, w, stride, pad, pad, stats=True)
    with env(MAAI_CONV_PP="2"):
        y1, st1 = K.conv2d(x, w, stride, pad, pad, stats=True)
        y2, st2 = K.conv2d(x, w, stride, pad, pad, stats=True)
        y3 = K.conv2d(x, w, stride, pad, pad)                      # without the statistics epilogue
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), "ping-pong kernel output differs from the ring kernel's"
    assert torch.equal(y1, y2) and torch.equal(st1, st2) and torch.equal(y1, y3), "ping-pong kernel is not deterministic"
    m = y1.numel() // cout
    assert st1.shape == ((m + _rows(cout) - 1) // _rows(cout), 2, cout)
    t0, t1 = st0.double().sum(0), st1.double().sum(0)
    tol = 2e-6 * (st0.double().abs().sum(0) + 1.0)
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())
    # and against fp64 torch directly (small cases): the kernel is not only consistent with its sibling
    if m * cout * k * k * cin <= 3e9:
        ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), None, stride, pad).permute(0, 2, 3, 1)
        np.testing.assert_allclose(y1.double().cpu().numpy(), ref.numpy(), rtol=1.0 / 128, atol=2e-2)
        np.testing.assert_allclose(t1[0].cpu().numpy(), ref.reshape(-1, cout).sum(0).numpy(), rtol=1e-3, atol=2e-2 * m ** 0.5)


DG_CASES = [
    # N, H, W, C (of dy = Cout of the forward layer), Cb (channels of the unit below = the gradient's output), k
    (8, 96, 97, 128, 256, 3),        # 291 tiles, the last ragged: the persistent form
    (2, 14, 14, 256, 256, 3),
    (3, 15, 15, 512, 256, 1),
    (1, 9, 11, 128, 256, 3),
    (2, 16, 16, 1024, 512, 1),
    (3, 23, 23, 128, 128, 3),        # the 4 x 2 wave grid
    (2, 16, 16, 512, 128, 1),
]


@pytest.mark.parametrize("mask", ["from_y", "bits", "tensor"])
@pytest.mark.parametrize("acc", [False, True], ids=["store", "accumulate"])
@pytest.mark.parametrize("case", DG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pingpong_data_gradient_epilogue_matches_ring_kernel(K, case, acc, mask):
    """dx = conv(dy, W^T) * [unit below's output > 0] (+ previous content), with the BatchNorm-backward partial sums of the
    stored gradient (MAAI_EPI_DGRAD_REDUCE): bit-identical dx, sums to fp32 summation order."""
    n, h, w_, c, cb, k = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 5)
    dy = (torch.randn(n, h, w_, c, generator=g) * 0.1).cuda().bfloat16()
    wq = (torch.randn(cb, k, k, c, generator=g) / (k * k * c) ** 0.5).cuda().bfloat16()
    yb = torch.randn(n, h, w_, cb, generator=g).cuda().bfloat16()       # the unit below's raw conv output
    mean = (torch.randn(cb, generator=g) * 0.1).cuda()
    s, t = (torch.rand(cb, generator=g) + 0.5).cuda(), (torch.randn(cb, generator=g) * 0.3).cuda()
    prev = torch.randn(n, h, w_, cb, generator=g).cuda().bfloat16()
    pad = k // 2
    out_act, bits = K.bn_act_fwd(yb, s, t, torch.randn(n, h, w_, cb, generator=g).cuda().bfloat16(), True, want_bits=True)
    res = []
    for mode in ("0", "2"):
        with env(MAAI_CONV_PP=mode):
            rows = K.conv2d_stats_rows(dy, wq, 1, pad, pad)
            slab = torch.zeros(rows, 2, cb, device="cuda")
            out = prev.clone() if acc else torch.empty_like(yb)
            kw = dict(accumulate=acc)
            if mask == "from_y":
                K.conv2d_store_reduce(dy, wq, 1, pad, pad, out, slab, yb, mean, s, t, None, **kw)
            elif mask == "bits":
                K.conv2d_store_reduce(dy, wq, 1, pad, pad, out, slab, yb, mean, None, None, bits, mask_bits=True, **kw)
            else:
                K.conv2d_store_reduce(dy, wq, 1, pad, pad, out, slab, yb, mean, None, None, out_act, **kw)
            res.append((out, slab))
    torch.cuda.synchronize()
    (o0, s0), (o1, s1) = res
    assert torch.equal(o0, o1)
    m = n * h * w_
    assert s1.shape[0] == (m + _rows(cb) - 1) // _rows(cb)
    t0, t1 = s0.double().sum(0), s1.double().sum(0)
    tol = 4e-6 * (s0.double().abs().sum(0) + 1.0) + 1e-5 * float(t0.abs().max())
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())


def test_shape_rule(K):
    """default: the 3x3 layers with >= 256 input channels; a forced ring-kernel tile wins; the slab has one row per 256
    pixels (the ring kernel: per 128 at this small size, per 256 at the benchmark's), as maai_conv2d_stats_rows says"""
    x = torch.randn(2, 14, 14, 256).cuda().bfloat16()
    w = (torch.randn(256, 3, 3, 256) / 48).cuda().bfloat16()
    with env(MAAI_CONV_PP="0"):
        y0, st0 = K.conv2d(x, w, 1, 1, 1, stats=True)
    y1, st1 = K.conv2d(x, w, 1, 1, 1, stats=True)                # default rule: ping-pong
    with env(MAAI_CONV_BM="128"):
        y2, st2 = K.conv2d(x, w, 1, 1, 1, stats=True)            # forced ring tile
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    assert st1.shape == (2, 2, 256) and st0.shape == st2.shape == (4, 2, 256)
    assert K.conv2d_stats_rows(x, w, 1, 1, 1) == 2
    with env(MAAI_CONV_PP="0"):
        assert K.conv2d_stats_rows(x, w, 1, 1, 1) == 4


WG_CASES = [
    # N, H, W, Cin, Cout, k, stride
    (4, 32, 32, 256, 256, 3, 1),      # M = 4096: 64 K-tiles
    (6, 28, 28, 512, 512, 3, 1),      # M = 4704: a ragged last K-tile; 2 x 18 tiles of dw
    (8, 28, 28, 256, 1024, 1, 1),     # dense pointwise, four row tiles
    (8, 28, 28, 1024, 256, 1, 1),     # dense pointwise, four column tiles
    (5, 57, 57, 256, 256, 3, 2),      # stride 2 on an odd extent (OH = 29)
    (16, 33, 35, 256, 512, 1, 2),     # strided pointwise (a projection shortcut)
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_pingpong_weight_gradient_matches_ring_kernels_and_fp64(K, case):
    """csrc/conv_ppw.hip (8-wave ping-pong, 256 x 256 tiles of dw, transposed fragment reads) against the weight-gradient
    kernels it replaces on these shapes (ring / patch, pinned to the oracle by test_gpu_kernels.py) and against fp64
    autograd of conv2d directly: same products, fp32 sums in a different order."""
    n, h, w_, cin, cout, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 9)
    pad = k // 2
    x = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    oh, ow = (h + 2 * pad - k) // stride + 1, (w_ + 2 * pad - k) // stride + 1
    dy = (torch.randn(n, oh, ow, cout, generator=g) * 0.05).cuda().bfloat16()
    K.AUTOTUNE[0] = False
    try:
        with env(MAAI_WGRAD_PP="0"):
            d0 = K.conv2d_wgrad(x, dy, k, k, stride, pad, pad)
        with env(MAAI_WGRAD_PP="2"):
            d1 = K.conv2d_wgrad(x, dy, k, k, stride, pad, pad)
            d2 = K.conv2d_wgrad(x, dy, k, k, stride, pad, pad)
    finally:
        K.AUTOTUNE[0] = True
    torch.cuda.synchronize()
    scale = float(d0.abs().max())
    assert d1.shape == d0.shape == (cout, k, k, cin)
    assert float((d1 - d0).abs().max()) <= 2e-5 * scale + 1e-6, float((d1 - d0).abs().max() / scale)
    assert float((d1 - d2).abs().max()) <= 2e-5 * scale + 1e-6      # (fp32 atomics: not bit-reproducible, documented)
    wz = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wz, None, stride, pad).backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref = wz.grad.permute(0, 2, 3, 1)
    np.testing.assert_allclose(d1.cpu().double().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))


WIDE_CASES = [
    # N, H, W, Cin, Cout, PW
    (4, 32, 32, 128, 128, 16),      # interior + border patches, one tile of dw
    (4, 32, 32, 128, 128, 32),
    (3, 28, 28, 512, 512, 32),      # the 512-channel layer's plane: 32-wide patches cover 87.5 %; 4 x 8 tiles of dw
    (3, 28, 28, 256, 128, 16),      # ragged on both axes
    (2, 56, 56, 256, 256, 16),      # the 256-channel layer's plane
    (2, 19, 37, 64, 128, 32),       # odd extents, 64 input channels
    (1, 5, 3, 64, 128, 16),         # a plane smaller than one patch
    # 64 output channels (or an odd multiple): the K-split form — the two wave groups share a patch's pixels
    (2, 32, 32, 64, 64, 16),
    (2, 30, 33, 64, 64, 32),
    (2, 19, 37, 128, 192, 32),
    (1, 5, 3, 64, 64, 16),
    # stride 2 (the 3x3 layers that open stages 2-4): 16 x 4 output patches, 9 x 33 input halos — PW < 0 marks them
    (3, 32, 32, 128, 128, -16),
    (2, 57, 57, 256, 256, -16),     # odd input extent (OH = 29): ragged patches on both axes
    (2, 28, 28, 64, 128, -16),
    (1, 7, 5, 64, 128, -16),        # a plane smaller than one patch
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wide_patch_weight_gradient_matches_patch_kernel_and_fp64(K, case):
    """csrc/conv_wgrad3w.hip (eight waves, 128 x 9 x 64 blocks of dw, both patch geometries) against the kernels it replaces
    (six-wave patch / ring, pinned to the oracle by test_gpu_kernels.py) and against fp64 autograd of conv2d
    (resnet.py:20-23): same products, fp32 sums in a different order; and it is the kernel that ran."""
    from maai_hip._lib import lib
    n, h, w_, cin, cout, pw = case
    stride = 2 if pw < 0 else 1
    pw = abs(pw)
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    oh, ow = (h - 1) // stride + 1, (w_ - 1) // stride + 1
    dy = (torch.randn(n, oh, ow, cout, generator=g) * 0.05).cuda().bfloat16()
    K.AUTOTUNE[0] = False
    were = lib().maai_kernel_names(1)
    try:
        with env(MAAI_WGRAD_WIDE="0", MAAI_WGRAD_PP="0"):
            d0 = K.conv2d_wgrad(x, dy, 3, 3, stride, 1, 1)
            assert b"wide" not in lib().maai_last_kernel_name()
        with env(MAAI_WGRAD_WIDE="2", MAAI_WGRAD_WIDE_PW=str(pw)):
            d1 = K.conv2d_wgrad(x, dy, 3, 3, stride, 1, 1)
            assert b"wgrad3x3_wide_kernel<%d, %s, %d>" % (pw, b"true" if cout % 128 else b"false", stride) in lib().maai_last_kernel_name()
            d2 = K.conv2d_wgrad(x, dy, 3, 3, stride, 1, 1, target_blocks=3072)   # a different pixel split
    finally:
        K.AUTOTUNE[0] = True
        lib().maai_kernel_names(were)
    torch.cuda.synchronize()
    scale = float(d0.abs().max())
    assert d1.shape == d0.shape == (cout, 3, 3, cin)
    assert float((d1 - d0).abs().max()) <= 2e-5 * scale + 1e-6, float((d1 - d0).abs().max() / scale)
    assert float((d1 - d2).abs().max()) <= 2e-5 * scale + 1e-6      # (fp32 atomics: not bit-reproducible, documented)
    wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wz, None, stride, 1).backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref = wz.grad.permute(0, 2, 3, 1)
    np.testing.assert_allclose(d1.cpu().double().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))
