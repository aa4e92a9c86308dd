"""The persistent resident-weights kernel for the 64 -> 64 channel 3x3 stride-1 layers (csrc/conv_c64.hip: weights in LDS
for the whole launch, one stream of 16 x 4-pixel patches per wave, next halo prefetched into registers, statistics
accumulated per wave) against the halo kernel of conv_igemm.h it replaces at 224^2 (pinned to the oracle / fp64 torch by
test_gpu_kernels.py and test_gpu_timed_size.py): same K order, so outputs must be BIT-IDENTICAL — plain and
normalise-on-load inputs, planes that the 16 x 4 patches do not tile, many patches per wave — statistics to fp32
summation order, and the data-gradient epilogue (every mask form + BatchNorm-backward sums) likewise.
Reference semantics: resnet.py:107 (conv2 = conv3x3(width, width) of a Bottleneck) and its gradient."""
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


HALO = dict(MAAI_CONV_C64="0", MAAI_CONV_HALO="1")   # the kernel it replaces (32-channel chunk major, tap minor)
C64 = dict(MAAI_CONV_C64="2")

# N, H, W
CASES = [
    (2, 32, 32),      # exact tiling, 32 patches: fewer patches than waves
    (3, 30, 30),      # W, H not multiples of 16 / 4: ragged right and bottom patches
    (1, 9, 11),       # smaller than one patch row
    (4, 64, 48),
    (16, 56, 56),     # 3136 patches on 1536 waves: two or three patches per wave (prefetch, halo reuse, sums across patches)
    (2, 224, 224),    # the benchmark's plane
    (2, 20, 20),      # W % 16 in 1..6: the last patch column ends inside the halo's FIRST column group (the
    (1, 33, 35),      # missing "< W" test loaded the next row's pixels — and ran past the tensor on the last row)
]


def _coeffs(c, g):
    s = ((torch.rand(c, generator=g) * 1.5 + 0.25) * torch.where(torch.rand(c, generator=g) < 0.2, -1.0, 1.0)).cuda()
    return s, (torch.randn(c, generator=g) * 0.7).cuda()


@pytest.mark.parametrize("mode", ["plain", "relu", "lin"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_forward_matches_halo_kernel(K, case, mode):
    n, h, w_ = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 3)
    x = torch.randn(n, h, w_, 64, generator=g).cuda().bfloat16()
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24).cuda().bfloat16()
    inp = x
    if mode != "plain":
        s, t = _coeffs(64, g)
        inp = K.Lazy(x, s, t, mode == "relu")
    with env(**HALO):
        y0, st0 = K.conv2d(inp, w, 1, 1, 1, stats=True)
    with env(**C64):
        assert K.conv2d_kernel_family(x, w, 1, 1, 1) == 3
        y1, st1 = K.conv2d(inp, w, 1, 1, 1, stats=True)
        y2, st2 = K.conv2d(inp, w, 1, 1, 1, stats=True)
        y3 = K.conv2d(inp, w, 1, 1, 1)
        rows = K.conv2d_stats_rows(x, w, 1, 1, 1)
        if mode != "plain":   # the lazy launch == BatchNorm pass + plain launch on the SAME kernel, slab included
            act = K.bn_act_fwd(x, inp.scale, inp.shift, None, inp.relu)
            y4, st4 = K.conv2d(act, w, 1, 1, 1, stats=True)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), "persistent kernel output differs from the halo kernel's"
    assert torch.equal(y1, y2) and torch.equal(st1, st2) and torch.equal(y1, y3), "not deterministic"
    assert st1.shape == (rows, 2, 64)
    if mode != "plain":
        assert torch.equal(y4, y1) and torch.equal(st4, st1)
    t0, t1 = st0.double().sum(0), st1.double().sum(0)
    tol = 2e-6 * (st0.double().abs().sum(0) + 1.0)
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())
    if mode == "plain" and n * h * w_ <= 20000:
        ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        np.testing.assert_allclose(y1.double().cpu().numpy(), ref.numpy(), rtol=1.0 / 128, atol=2e-2)
        np.testing.assert_allclose(t1[0].cpu().numpy(), ref.reshape(-1, 64).sum(0).numpy(), rtol=1e-3, atol=2e-2 * (n * h * w_) ** 0.5)


@pytest.mark.parametrize("mask", ["from_y", "bits", "tensor", "none"])
@pytest.mark.parametrize("case", [(2, 32, 32), (3, 30, 30), (16, 56, 56), (2, 20, 20), (1, 33, 35)], ids=lambda c: "x".join(map(str, c)))
def test_data_gradient_epilogue_matches_halo_kernel(K, case, mask):
    """dx = conv(dy, W^T) * [unit below's output > 0] with the BatchNorm-backward partial sums of the stored gradient."""
    n, h, w_ = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 4)
    dy = (torch.randn(n, h, w_, 64, generator=g) * 0.1).cuda().bfloat16()
    wq = (torch.randn(64, 3, 3, 64, generator=g) / 24).cuda().bfloat16()
    yb = torch.randn(n, h, w_, 64, generator=g).cuda().bfloat16()
    mean = (torch.randn(64, generator=g) * 0.1).cuda()
    s, t = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.3).cuda()
    out_act, bits = K.bn_act_fwd(yb, s, t, torch.randn(n, h, w_, 64, generator=g).cuda().bfloat16(), True, want_bits=True)
    res = []
    for ev in (HALO, C64):
        with env(**ev):
            rows = K.conv2d_stats_rows(dy, wq, 1, 1, 1)
            slab = torch.zeros(rows, 2, 64, device="cuda")
            out = torch.empty_like(yb)
            if mask == "from_y":
                K.conv2d_store_reduce(dy, wq, 1, 1, 1, out, slab, yb, mean, s, t, None)
            elif mask == "bits":
                K.conv2d_store_reduce(dy, wq, 1, 1, 1, out, slab, yb, mean, None, None, bits, mask_bits=True)
            elif mask == "tensor":
                K.conv2d_store_reduce(dy, wq, 1, 1, 1, out, slab, yb, mean, None, None, out_act)
            else:
                K.conv2d_store_reduce(dy, wq, 1, 1, 1, out, slab, yb, mean, None, None, None)
            res.append((out, slab))
    torch.cuda.synchronize()
    (o0, s0), (o1, s1) = res
    assert torch.equal(o0, o1)
    t0, t1 = s0.double().sum(0), s1.double().sum(0)
    tol = 4e-6 * (s0.double().abs().sum(0) + 1.0) + 1e-5 * float(t0.abs().max())
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())


def test_shape_rule(K):
    """default: large planes only (the benchmark's 224^2 x 256 images); small inputs stay on the ring / halo kernels"""
    w = (torch.randn(64, 3, 3, 64) / 24).cuda().bfloat16()
    small = torch.randn(2, 32, 32, 64).cuda().bfloat16()
    assert K.conv2d_kernel_family(small, w, 1, 1, 1) == 0
    big = torch.empty(64, 224, 224, 64, dtype=torch.bfloat16, device="cuda")
    assert K.conv2d_kernel_family(big, w, 1, 1, 1) == 3
    with env(MAAI_CONV_C64="0"):
        assert K.conv2d_kernel_family(big, w, 1, 1, 1) == 0
