"""Normalise-on-load (maai_conv_epilogue.xs/xt/xb, maai_conv2d_wgrad_xf): a convolution / weight gradient whose input
is the RAW output of the unit below must give, bit for bit, what the materialised path gives — maai_bn_act_fwd (or
_fwd2 / the residual form) followed by the plain launch — on every tile / staging variant, including zero padding
(applied to the activation, not to the raw tensor), ragged planes, strides and rows beyond M.  The materialised path
itself is pinned to the oracle in test_gpu_kernels.py.  Reference semantics: resnet.py:101-109,126-133."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from maai_hip import kernels
    return kernels


@pytest.fixture(autouse=True)
def _ring_family():
    """These cases compare the normalise-on-load launches with the materialised path ON THE SAME KERNEL FAMILY (ring /
    halo / streaming): the statistics slab is compared bit for bit, and each family sums it in its own order.  The 8-wave
    ping-pong kernel (conv_pp.hip) takes tensors only — the engine materialises a lazy input of a layer that runs on it
    (engine._unit_fwd_gen), so a mixed pair never occurs in the product; tests/test_gpu_pp.py covers that kernel."""
    old = os.environ.get("MAAI_CONV_PP")
    os.environ["MAAI_CONV_PP"] = "0"
    yield
    if old is None:
        os.environ.pop("MAAI_CONV_PP", None)
    else:
        os.environ["MAAI_CONV_PP"] = old


class env(object):
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _coeffs(c, g):
    scale = (torch.rand(c, generator=g) * 1.5 + 0.25) * torch.where(torch.rand(c, generator=g) < 0.2, -1.0, 1.0)
    shift = torch.randn(c, generator=g) * 0.7
    return scale.cuda(), shift.cuda()


# N, Cin, H, W, Cout, k, stride, pad
CASES = [
    (2, 64, 30, 30, 256, 1, 1, 0),      # conv3 of a bottleneck (K = 64: two-slot ring)
    (3, 256, 15, 15, 64, 1, 1, 0),      # conv1
    (2, 128, 9, 11, 512, 1, 1, 0),      # M = 198: rows beyond M in the last tile must stay zero (statistics)
    (2, 64, 30, 30, 64, 3, 1, 1),       # conv2: halo / row staged
    (2, 128, 30, 30, 128, 3, 2, 1),     # conv2 of a stage's first block, stride 2
    (3, 128, 15, 15, 128, 3, 2, 1),     # odd extent
    (2, 256, 15, 15, 512, 1, 2, 0),     # 1x1 stride 2 (not the pointwise fast path)
    (5, 64, 9, 7, 192, 3, 1, 1),        # ragged everything
    (1, 512, 4, 4, 2048, 1, 1, 0),      # M = 16 << tile
    (2, 256, 12, 12, 1024, 1, 1, 0),    # expanding layer (128x256 tile when forced)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("relu", [True, False], ids=["relu", "lin"])
@pytest.mark.parametrize("variant", ["default", "bm256", "halo", "rows", "bn256"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_lazy_input_bit_identical(K, case, variant, relu, dtype):
    n, cin, h, w, cout, k, stride, pad = case
    if dtype == torch.float32 and variant != "default":
        pytest.skip("fp32 storage uses the 128-row tile only")
    if variant in ("halo", "rows") and not (k == 3 and stride == 1):
        pytest.skip("halo staging is for 3x3 stride-1 layers")
    if variant == "bn256" and not (k == 1 and cout % 256 == 0):
        pytest.skip("256-column tile is for pointwise layers")
    if not relu and variant != "default":
        pytest.skip("covered by the default variant")
    ev = {"default": {}, "bm256": {"MAAI_CONV_BM": "256"}, "halo": {"MAAI_CONV_HALO": "1"}, "rows": {"MAAI_CONV_HALO": "0"},
          "bn256": {"MAAI_CONV_BN": "256"}}[variant]
    g = torch.Generator().manual_seed(1000 + cin + h)
    y = (torch.randn(n, h, w, cin, generator=g) * 1.3).to(dtype).cuda()
    wt = (torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5).to(dtype).cuda()
    scale, shift = _coeffs(cin, g)
    with env(**ev):
        act = K.bn_act_fwd(y, scale, shift, None, relu)
        ref, ref_part = K.conv2d(act, wt, stride, pad, pad, stats=True)
        got, got_part = K.conv2d(K.Lazy(y, scale, shift, relu), wt, stride, pad, pad, stats=True)
        torch.cuda.synchronize()
    assert got_part.shape == ref_part.shape
    assert torch.equal(got, ref)
    assert torch.equal(got_part, ref_part)
    assert float(ref.float().abs().max()) > 0.1


JOIN_CASES = [
    (2, 256, 30, 30, 64),      # next block's conv1 at layer-1 width
    (3, 512, 9, 11, 128),      # ragged M
    (2, 1024, 8, 8, 256),      # two column tiles: the joined activation must be written exactly once
    (1, 2048, 4, 4, 512),
    (2, 64, 16, 16, 256),      # K = 64
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("two_bn", [False, True], ids=["identity", "downsample"])
@pytest.mark.parametrize("variant", ["default", "bm256", "bn256"])
@pytest.mark.parametrize("case", JOIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_lazy_join_bit_identical(K, case, variant, two_bn, dtype):
    """relu(bn3(y3) + shortcut) formed on load by the next block's conv1 (resnet.py:126-133): output, statistics,
    the joined activation handed back for the shortcut, and its 1-bit ReLU mask."""
    n, cin, h, w, cout = case
    if dtype == torch.float32 and variant != "default":
        pytest.skip("fp32 storage uses the 128-row tile only")
    if variant == "bn256" and cout % 256:
        pytest.skip("needs Cout % 256 == 0")
    ev = {"default": {}, "bm256": {"MAAI_CONV_BM": "256"}, "bn256": {"MAAI_CONV_BN": "256"}}[variant]
    g = torch.Generator().manual_seed(77 + cin)
    y = (torch.randn(n, h, w, cin, generator=g) * 1.3).to(dtype).cuda()
    b = (torch.randn(n, h, w, cin, generator=g)).to(dtype).cuda()
    if not two_bn:
        b = b.clamp_min(0)   # an identity shortcut is a post-ReLU tensor
    wt = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).to(dtype).cuda()
    s1, t1 = _coeffs(cin, g)
    s2, t2 = _coeffs(cin, g)
    bits_ok = dtype == torch.bfloat16
    with env(**ev):
        if two_bn:
            r = K.bn_act_fwd2(y, s1, t1, b, s2, t2, True, want_bits=bits_ok)
            lazy = K.Lazy(y, s1, t1, True, b, s2, t2)
        else:
            r = K.bn_act_fwd(y, s1, t1, b, True, want_bits=bits_ok)
            lazy = K.Lazy(y, s1, t1, True, b)
        act, bits = r if bits_ok else (r, None)
        ref, ref_part = K.conv2d(act, wt, stats=True)
        sentinel = None
        res = K.conv2d(lazy, wt, stats=True, join_out=True, join_bits=bits_ok)
        torch.cuda.synchronize()
    got, got_part, joined = res[0], res[1], res[2]
    assert torch.equal(joined, act)
    if bits_ok:
        assert torch.equal(res[3], bits)
    assert torch.equal(got, ref)
    assert torch.equal(got_part, ref_part)
    # without the hand-back the convolution is the same
    with env(**ev):
        only = K.conv2d(lazy, wt)
    assert torch.equal(only, ref)


WGRAD_CASES = [
    (2, 64, 30, 30, 64, 3, 1, 1),       # patch kernel when forced, ring otherwise
    (2, 64, 30, 30, 256, 1, 1, 0),
    (3, 128, 15, 15, 128, 3, 2, 1),
    (2, 256, 15, 15, 512, 1, 2, 0),
    (5, 64, 9, 7, 192, 3, 1, 1),
    (2, 128, 17, 33, 64, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("patch", ["0", "1", "1s"], ids=["ring", "patch", "patch-single-buffered"])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_lazy_input_exact(K, case, patch, dtype):
    """Weight gradient with the activation formed on load.  Integer data (every product and partial sum exact in
    fp32, whatever the order of the atomic adds): bit-identical to the materialised path and to fp64 autograd."""
    n, cin, h, w, cout, k, stride, pad = case
    if patch != "0" and not (k == 3 and stride == 1 and dtype == torch.bfloat16):
        pytest.skip("patch-staged kernel: bf16 3x3 stride 1")
    g = torch.Generator().manual_seed(5 + cin + h)
    y = torch.randint(-6, 7, (n, h, w, cin), generator=g).float()
    scale = torch.tensor([0.5, 1.0, 2.0, -1.0])[torch.randint(0, 4, (cin,), generator=g)]
    shift = torch.randint(-3, 4, (cin,), generator=g).float()
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randint(-2, 3, (n, oh, ow, cout), generator=g).float()
    yd, dyd, sc, sh = y.to(dtype).cuda(), dy.to(dtype).cuda(), scale.cuda(), shift.cuda()
    with env(MAAI_WGRAD_PATCH=patch[0], MAAI_WGRAD_PATCH_DB="0" if patch == "1s" else "1", MAAI_AUTOTUNE="0"):
        K.AUTOTUNE[0] = False
        try:
            act = K.bn_act_fwd(yd, sc, sh, None, True)
            ref = K.conv2d_wgrad(act, dyd, k, k, stride, pad, pad)
            got = K.conv2d_wgrad(K.Lazy(yd, sc, sh, True), dyd, k, k, stride, pad, pad)
            torch.cuda.synchronize()
        finally:
            K.AUTOTUNE[0] = True
    assert torch.equal(got, ref)
    a64 = torch.relu(y.double() * scale.double() + shift.double()).permute(0, 3, 1, 2)
    wz = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    out = torch.nn.functional.conv2d(a64, wz, None, stride, pad)
    out.backward(dy.double().permute(0, 3, 1, 2))
    np.testing.assert_array_equal(got.cpu().permute(0, 3, 1, 2).double().numpy(), wz.grad.numpy())


def test_wgrad_lazy_random_data(K):
    """Random (non-integer) data: same values up to the order of the fp32 atomic adds."""
    g = torch.Generator().manual_seed(9)
    for (n, cin, h, w, cout, k, stride, pad) in [(4, 64, 28, 28, 64, 3, 1, 1), (4, 128, 14, 14, 512, 1, 1, 0)]:
        y = torch.randn(n, h, w, cin, generator=g).bfloat16().cuda()
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        dy = torch.randn(n, oh, ow, cout, generator=g).bfloat16().cuda()
        sc, sh = _coeffs(cin, g)
        act = K.bn_act_fwd(y, sc, sh, None, True)
        ref = K.conv2d_wgrad(act, dy, k, k, stride, pad, pad)
        got = K.conv2d_wgrad(K.Lazy(y, sc, sh, True), dy, k, k, stride, pad, pad)
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-4 * float(ref.abs().max()))
