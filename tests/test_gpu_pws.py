"""The streaming pointwise kernel (csrc/conv_pws.hip: A operand in registers, weights streamed through LDS, all column
tiles per workgroup) against the ring kernel (csrc/conv_igemm.h), which test_gpu_kernels.py pins to the oracle: the same
MFMA sequence per output element, so outputs must be bit-identical — plain and normalise-on-load inputs, both column
tiles, rows beyond M, one to many column tiles — and the BatchNorm statistics slab (one row per 128 pixels in both) must
agree to fp32 summation order.  Reference semantics: resnet.py:101-109 (conv1 / conv3 of a bottleneck)."""
import os

import pytest
import torch

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


# N, H, W, Cin, Cout
CASES = [
    (2, 30, 30, 64, 256),     # conv3 of layer 1 (two K-steps per column tile: the densest epilogue / prefetch interleave)
    (2, 30, 30, 64, 64),      # ONE column tile, fewer weight stages than ring slots
    (3, 15, 15, 128, 512),    # M = 675: rows beyond M in the last tile
    (1, 9, 11, 128, 128),     # M = 99 < one tile
    (2, 16, 16, 256, 1024),   # layer 3: eight K-steps, eight / sixteen column tiles
    (2, 12, 12, 256, 64),     # channel-reducing (kernel is built for it; the shape rule keeps the ring kernel)
    (5, 7, 9, 256, 192),      # Cout not a multiple of 128
    (1, 56, 56, 128, 512),    # many row tiles
]


def _run(K, x, w, lazy):
    return K.conv2d(K.Lazy(*lazy) if lazy else x, w, 1, 0, 0, stats=True)


@pytest.mark.parametrize("bn", ["64", "128"])
@pytest.mark.parametrize("mode", ["plain", "relu", "lin"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_streaming_pointwise_matches_ring_kernel(K, case, mode, bn):
    n, h, w_, cin, cout = case
    g = torch.Generator().manual_seed(hash(case) % 10007)
    x = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).cuda().bfloat16()
    lazy = None
    if mode != "plain":
        s = ((torch.rand(cin, generator=g) * 1.5 + 0.25) * torch.where(torch.rand(cin, generator=g) < 0.2, -1.0, 1.0)).cuda()
        t = (torch.randn(cin, generator=g) * 0.7).cuda()
        lazy = (x, s, t, mode == "relu")
    with env(MAAI_CONV_PWS="0"):
        y0, st0 = _run(K, x, w, lazy)
    with env(MAAI_CONV_PWS="2", MAAI_PWS_BN=bn):
        y1, st1 = _run(K, x, w, lazy)
        y2, st2 = _run(K, x, w, lazy)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), "streaming kernel output differs from the ring kernel's"
    assert torch.equal(y1, y2) and torch.equal(st1, st2), "streaming kernel is not deterministic"
    m = n * h * w_
    assert st1.shape == ((m + 127) // 128, 2, cout)
    t0, t1 = st0.double().sum(0), st1.double().sum(0)
    # sums of ~m fp32 terms in two different orders
    tol = 2e-6 * (st0.double().abs().sum(0) + 1.0)
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())
    # the slab's rows are per-tile sums of the (fp32) accumulators: check one tile against the stored bf16 output
    rows = min(128, m)
    ref = y1.reshape(m, cout)[:rows].double().sum(0)
    assert torch.allclose(st1[0, 0].double(), ref, rtol=0, atol=0.02 * rows ** 0.5 + 0.01 * float(ref.abs().max()))


def test_shape_rule_and_knob_precedence(K):
    """expanding layers take the streaming kernel by default (slab rows = M/128 either way), a forced ring tile wins"""
    x = torch.randn(2, 32, 32, 256).cuda().bfloat16()
    w = (torch.randn(1024, 1, 1, 256) / 16).cuda().bfloat16()
    with env(MAAI_CONV_PWS="0"):
        y0, st0 = K.conv2d(x, w, stats=True)
    y1, st1 = K.conv2d(x, w, stats=True)                    # default: streaming
    with env(MAAI_CONV_BN="256"):
        y2, st2 = K.conv2d(x, w, stats=True)                # forced 128x256 ring tile
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    assert st0.shape == st1.shape == st2.shape == (16, 2, 1024)
    assert torch.equal(st0, st2) and not torch.equal(st0, st1)   # the two kernels sum the rows of a tile in different orders


@pytest.mark.parametrize("bn", ["64", "128"])
@pytest.mark.parametrize("shortcut", ["identity", "projection"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256), (3, 15, 15, 256, 64), (1, 9, 11, 128, 128), (2, 12, 12, 256, 1024)],
                         ids=lambda c: "x".join(map(str, c)))
def test_streaming_join_on_load_matches_ring_kernel(K, case, shortcut, bn):
    """the residual join relu(bn3(y3) + shortcut) formed on load (resnet.py:126-133): output, joined tensor, 1-bit mask"""
    n, h, w_, cin, cout = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 1)
    y3 = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    sc = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).cuda().bfloat16()
    s, t = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.5).cuda()
    if shortcut == "projection":
        s2, t2 = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.5).cuda()
        lazy = K.Lazy(y3, s, t, True, sc, s2, t2)
    else:
        lazy = K.Lazy(y3, s, t, True, sc.clamp_min(0))
    res = []
    for kv in ({"MAAI_CONV_PWS": "0"}, {"MAAI_CONV_PWS": "2", "MAAI_PWS_BN": bn}):
        with env(**kv):
            res.append(K.conv2d(lazy, w, stats=True, join_out=True, join_bits=True))
    torch.cuda.synchronize()
    (y0, st0, j0, b0), (y1, st1, j1, b1) = res
    assert torch.equal(y0, y1) and torch.equal(j0, j1) and torch.equal(b0, b1)
    m = n * h * w_
    assert st1.shape == ((m + 127) // 128, 2, cout)
    t0, t1 = st0.double().sum(0), st1.double().sum(0)
    assert bool(((t0 - t1).abs() <= 2e-6 * (st0.double().abs().sum(0) + 1.0)).all())


@pytest.mark.parametrize("keep", [False, True], ids=["nokeep", "keep-y"])
@pytest.mark.parametrize("shortcut", ["identity", "projection"])
@pytest.mark.parametrize("pre_lazy", [False, True], ids=["tensor", "lazy"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64), (3, 15, 15, 128), (1, 9, 11, 64), (1, 56, 56, 128)], ids=lambda c: "x".join(map(str, c)))
def test_chained_block_boundary_matches_unchained(K, case, pre_lazy, shortcut, keep):
    """conv3 -> bn3 -> (+ shortcut) -> relu -> next conv1 (resnet.py:118-133, :101) in ONE launch with conv3 recomputed
    (csrc/conv_chain.hip) against the sequence it replaces: conv3 stored, then the join-on-load conv1.  Bit-identical
    output, joined activation, mask and (when kept) recomputed tensor; the statistics-only launch gives conv3's slab."""
    n, h, w_, cout = case
    g = torch.Generator().manual_seed(hash(case) % 10007 + 3)
    a2 = torch.randn(n, h, w_, 64, generator=g).cuda().bfloat16()
    w3 = (torch.randn(256, 1, 1, 64, generator=g) / 8).cuda().bfloat16()
    w1 = (torch.randn(cout, 1, 1, 256, generator=g) / 16).cuda().bfloat16()
    sc = torch.randn(n, h, w_, 256, generator=g).cuda().bfloat16()
    s3, t3 = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.5).cuda()
    if pre_lazy:
        ps = ((torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.2, -1.0, 1.0)).cuda()
        pt = (torch.randn(64, generator=g) * 0.5).cuda()
        src = K.Lazy(a2, ps, pt, True)
    else:
        src = a2
    if shortcut == "projection":
        s2, t2 = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.5).cuda()
        extra = (sc, s2, t2)
    else:
        extra = (sc.clamp_min(0),)
    with env(MAAI_CONV_PWS="2", MAAI_PWS_BN="64"):   # (the reference join on the streaming kernel: same slab arithmetic)
        y3, st3 = K.conv2d(src, w3, stats=True)
        y1, st1, jo, jb = K.conv2d(K.Lazy(y3, s3, t3, True, *extra), w1, stats=True, join_out=True, join_bits=True)
    st3b = K.conv2d_stats_only(src, w3)
    got = K.conv2d_chained(K.Lazy(None, s3, t3, True, *extra, pre=(src, w3)), w1, stats=True, join_bits=True, keep_y=keep)
    torch.cuda.synchronize()
    assert torch.equal(st3, st3b)
    assert torch.equal(got[0], y1) and torch.equal(got[2], jo) and torch.equal(got[3], jb)
    assert torch.equal(got[1], st1)
    if keep:
        assert torch.equal(got[4], y3)
    y1r, st1r, jor, jbr = K.conv2d(K.Lazy(y3, s3, t3, True, *extra), w1, stats=True, join_out=True, join_bits=True)   # default dispatch
    assert torch.equal(got[0], y1r) and torch.equal(got[2], jor) and torch.equal(got[3], jbr)


@pytest.mark.parametrize("acc", [False, True], ids=["store", "accumulate"])
@pytest.mark.parametrize("case", [(2, 16, 16), (3, 15, 15), (1, 9, 11), (1, 56, 56), (4, 30, 30), (4, 224, 224), (3, 223, 225)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fused_conv3_backward_matches_the_two_launches(K, case, acc):
    """conv3 + bn3 backward of a 64 -> 256 bottleneck in one persistent launch (csrc/conv_bwd3.hip: BatchNorm-backward
    apply, data gradient with bn2's mask and sums, weight gradient; dz3 never written) against the sequence it replaces —
    the apply-on-load data gradient that keeps dz3, then the weight gradient on the lazy input (resnet.py:118-119 backward).
    The 4x224x224 / 3x223x225 cases have 3136 / 2352 tiles for 512 persistent workgroups: every workgroup walks several
    tiles (the top-of-tile barrier, the C area aliasing the a2 image on later tiles, dx accumulated across tiles, a ragged
    last tile); tests/test_gpu_timed_size.py runs the 256-image size."""
    n, h, w_ = case
    g_ = torch.Generator().manual_seed(hash(case) % 10007 + 11)
    g = torch.randn(n, h, w_, 256, generator=g_).cuda().bfloat16()
    y3 = torch.randn(n, h, w_, 256, generator=g_).cuda().bfloat16()
    y2 = torch.randn(n, h, w_, 64, generator=g_).cuda().bfloat16()
    wd = (torch.randn(64, 1, 1, 256, generator=g_) / 16).cuda().bfloat16()
    k1, k2, k3 = (torch.rand(256, generator=g_) + 0.5).cuda(), (torch.randn(256, generator=g_) * 0.1).cuda(), (torch.randn(256, generator=g_) * 0.1).cuda()
    mean2 = (torch.randn(64, generator=g_) * 0.1).cuda()
    s2, t2 = (torch.rand(64, generator=g_) + 0.5).cuda(), (torch.randn(64, generator=g_) * 0.3).cuda()
    m = n * h * w_
    prev = torch.randn(n, h, w_, 64, generator=g_).cuda().bfloat16()   # (accumulate: the other branch's gradient)
    dx0 = prev.clone() if acc else torch.empty_like(y2)
    part0 = torch.zeros((m + 127) // 128, 2, 64, device="cuda")
    dz = torch.empty_like(g)
    K.conv2d_store_reduce(g, wd, 1, 0, 0, dx0, part0, y2, mean2, s2, t2, None, accumulate=acc, axf=(y3, k1, k2, k3, dz))
    dw0 = K.conv2d_wgrad(K.Lazy(y2, s2, t2, True), dz, 1, 1, 1, 0, 0)
    dx1, part1, dw1 = K.conv_bwd3(g, y3, y2, wd, k1, k2, k3, mean2, s2, t2, dx=prev.clone() if acc else None)
    dx2, part2, dw2 = K.conv_bwd3(g, y3, y2, wd, k1, k2, k3, mean2, s2, t2, dx=prev.clone() if acc else None)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1) and torch.equal(dx1, dx2)
    t0, t1 = part0.double().sum(0), part1.double().sum(0)
    tol = 4e-6 * (part0.double().abs().sum(0) + 1.0) + 1e-5 * float(t0.abs().max())
    assert bool(((t0 - t1).abs() <= tol).all()), float(((t0 - t1).abs() / tol).max())
    assert dw1.shape == dw0.shape == (256, 1, 1, 64)
    scale = float(dw0.abs().max())
    assert float((dw1 - dw0).abs().max()) <= 2e-5 * scale + 1e-6 * m ** 0.5, float((dw1 - dw0).abs().max() / scale)
    a2 = K.bn_act_fwd(y2, s2, t2, None, True)
    ref = torch.einsum("mc,mk->ck", dz.reshape(m, 256).double(), a2.reshape(m, 64).double())
    assert float((dw1.reshape(256, 64).double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6 * m ** 0.5
