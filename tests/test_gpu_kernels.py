"""Kernel-level parity on a real MI355X: every entry point of the C ABI against
the CPU oracle / a plain fp64 torch reference on the same seeded inputs.
Tolerances: bf16 storage = 2^-8 relative on outputs that are rounded to bf16
(inputs are pre-rounded to bf16, products exact, fp32 accumulate); f32 storage
(exact-fp32 MFMA) = 2e-5 relative; integer/u8 work bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import simclr_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from maai_hip import kernels
    return kernels


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def khwc(w, dtype):
    return w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def tol(dtype):
    return dict(rtol=1.0 / 128, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=3e-5, atol=3e-5)


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 30, 30, 256, 1, 1, 0),
    (3, 256, 15, 15, 64, 1, 1, 0),
    (2, 64, 30, 30, 64, 3, 1, 1),
    (2, 128, 30, 30, 128, 3, 2, 1),
    (3, 128, 15, 15, 128, 3, 2, 1),     # odd extent, stride 2 (15 -> 8)
    (2, 256, 15, 15, 512, 1, 2, 0),     # downsample 1x1 stride 2
    (2, 32, 30, 30, 64, 7, 1, 3),       # padded-channel 7x7 stem
    (1, 512, 4, 4, 2048, 1, 1, 0),      # M = 16 << tile
    (5, 64, 9, 7, 192, 3, 1, 1),        # ragged everything, Cout = 3*64
]


@pytest.fixture(params=["128", "256"])
def conv_bm(request):
    """Row count of the implicit-GEMM tile (256-row tiles are normally chosen by shape; forced here so the small
    parity cases cover them, including the two-phase C-tile drain and the 4x1 wave column for 64-channel outputs)."""
    old = os.environ.get("MAAI_CONV_BM")
    os.environ["MAAI_CONV_BM"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("MAAI_CONV_BM", None)
    else:
        os.environ["MAAI_CONV_BM"] = old


@pytest.fixture(params=["0", "1"], ids=["rows", "halo"])
def conv_halo(request):
    """Halo-staged 16x16 patches for 3x3 stride-1 layers (normally chosen by shape; forced on / off here)."""
    old = os.environ.get("MAAI_CONV_HALO")
    os.environ["MAAI_CONV_HALO"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("MAAI_CONV_HALO", None)
    else:
        os.environ["MAAI_CONV_HALO"] = old


@pytest.mark.parametrize("case", [(2, 64, 30, 30, 64), (1, 64, 16, 16, 128), (3, 128, 13, 11, 128), (2, 32, 40, 24, 192),
                                  (1, 256, 33, 17, 256), (5, 64, 9, 7, 64)])
def test_conv3x3_halo_patches(K, case):
    """The halo-staged kernel on full, ragged and tiny planes: output and BatchNorm partial sums against fp64,
    and bit-for-bit against the row-staged kernel on integer data (exact arithmetic in any summation order)."""
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(77 + h)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    xi = torch.randint(-3, 4, (n, cin, h, w), generator=g).float()
    wi = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).float()
    got = {}
    old = os.environ.get("MAAI_CONV_HALO")
    try:
        for mode in ("0", "1"):
            os.environ["MAAI_CONV_HALO"] = mode
            y, part = K.conv2d(nhwc(x, torch.bfloat16), khwc(wt, torch.bfloat16), 1, 1, 1, stats=True)
            yi = K.conv2d(nhwc(xi, torch.bfloat16), khwc(wi, torch.bfloat16), 1, 1, 1)
            torch.cuda.synchronize()
            got[mode] = (from_nhwc(y), K.reduce_partials(part).cpu(), from_nhwc(yi))
    finally:
        if old is None:
            os.environ.pop("MAAI_CONV_HALO", None)
        else:
            os.environ["MAAI_CONV_HALO"] = old
    m = n * h * w
    for mode in ("0", "1"):
        y, sums, yi = got[mode]
        np.testing.assert_allclose(y.numpy(), ref.float().numpy(), **tol(torch.bfloat16))
        np.testing.assert_allclose(sums[:cout].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=2e-3 * m ** 0.5)
        np.testing.assert_allclose(sums[cout:].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=1e-3)
        assert torch.equal(yi, rb(F.conv2d(xi, wi, None, 1, 1)))
    assert torch.equal(got["0"][2], got["1"][2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_and_stats(K, case, dtype, conv_bm):
    n, cin, h, w, cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    if dtype == torch.bfloat16:
        x, wt = rb(x), rb(wt)
    ref = F.conv2d(x.double(), wt.double(), None, s, p)
    y, part = K.conv2d(nhwc(x, dtype), khwc(wt, dtype), s, p, p, stats=True)
    torch.cuda.synchronize()
    got = from_nhwc(y)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.float().numpy(), **tol(dtype))
    sums = K.reduce_partials(part).cpu()
    m = n * ref.shape[2] * ref.shape[3]
    np.testing.assert_allclose(sums[:cout].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=2e-3 * m ** 0.5)
    np.testing.assert_allclose(sums[cout:].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=1e-3)


def test_conv_exact_integer_layout(K, conv_bm):
    """Integer-valued operands: every product and sum is exact in bf16/fp32, so any
    fragment-layout or swizzle mistake shows as a hard mismatch (asymmetric data)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (2, 64, 13, 11), generator=g).float()
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float()
    ref = F.conv2d(x, w, None, 1, 1)
    for dtype in (torch.bfloat16, torch.float32):
        y = K.conv2d(nhwc(x, dtype), khwc(w, dtype), 1, 1, 1)
        got = from_nhwc(y)
        if dtype == torch.float32:
            assert torch.equal(got, ref)
        else:
            assert torch.equal(got, rb(ref))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_conv_scatter_accumulate(K, dtype, conv_bm):
    """strided scatter + read-modify-write epilogue (used by stride-2 data gradients)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 8, 8, generator=g)
    wt = torch.randn(128, 64, 1, 1, generator=g) / 8
    base = torch.randn(2, 128, 15, 15, generator=g)
    if dtype == torch.bfloat16:
        x, wt, base = rb(x), rb(wt), rb(base)
    out = nhwc(base, dtype)
    K.conv2d(nhwc(x, dtype), khwc(wt, dtype), 1, 0, 0, out=out, grid_hw=(8, 8), out_hw=(15, 15), out_stride=2, out_off=(0, 0),
             accumulate=True)
    ref = base.clone().double()
    ref[:, :, 0::2, 0::2] += F.conv2d(x.double(), wt.double())
    np.testing.assert_allclose(from_nhwc(out).numpy(), ref.float().numpy(), **tol(dtype))


WGRAD_CASES = [
    (2, 64, 30, 30, 256, 1, 1, 0),
    (2, 64, 30, 30, 64, 3, 1, 1),
    (3, 128, 15, 15, 128, 3, 2, 1),
    (2, 256, 15, 15, 512, 1, 2, 0),
    (2, 32, 30, 30, 64, 7, 1, 3),
    (4, 512, 4, 4, 128, 1, 1, 0),
    (3, 64, 9, 7, 192, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(K, case, dtype):
    n, cin, h, w, cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 997)
    x = torch.randn(n, cin, h, w, generator=g)
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = torch.randn(n, cout, oh, ow, generator=g)
    if dtype == torch.bfloat16:
        x, dy = rb(x), rb(dy)
    wt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, s, p).backward(dy.double())
    dw = K.conv2d_wgrad(nhwc(x, dtype), nhwc(dy, dtype), k, k, s, p, p)
    got = dw.cpu().permute(0, 3, 1, 2)
    scale = wt.grad.abs().max().item()
    np.testing.assert_allclose(got.numpy(), wt.grad.float().numpy(), rtol=1e-4, atol=2e-5 * scale + 1e-5)


def test_wgrad_exact_integer_layout(K):
    g = torch.Generator().manual_seed(9)
    x = torch.randint(-2, 3, (2, 64, 6, 5), generator=g).float()
    dy = torch.randint(-2, 3, (2, 128, 6, 5), generator=g).float()
    wt = torch.zeros(128, 64, 3, 3, requires_grad=True)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    for dtype in (torch.bfloat16, torch.float32):
        dw = K.conv2d_wgrad(nhwc(x, dtype), nhwc(dy, dtype), 3, 3, 1, 1, 1)
        assert torch.equal(dw.cpu().permute(0, 3, 1, 2), wt.grad)


@pytest.mark.parametrize("case", [(2, 64, 30, 30, 64), (1, 128, 16, 16, 64), (3, 64, 13, 11, 128), (2, 128, 40, 24, 192),
                                  (1, 64, 8, 16, 64), (5, 192, 9, 7, 64)])
@pytest.mark.parametrize("db", ["1", "0"], ids=["double-buffered", "single-buffered"])
def test_wgrad3x3_patch_kernel(K, case, db):
    """The patch-staged 3x3 weight gradient (normally chosen by shape, forced on here; both buffering schemes):
    against autograd in fp64 on random bf16 data, and bit-for-bit on integer data (exact in any order) — full, ragged
    and tiny planes, several (co, ci) tiles, more splits than patches."""
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(55 + h + cin)
    x, dy = rb(torch.randn(n, cin, h, w, generator=g)), rb(torch.randn(n, cout, h, w, generator=g))
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, 1, 1).backward(dy.double())
    xi = torch.randint(-2, 3, (n, cin, h, w), generator=g).float()
    dyi = torch.randint(-2, 3, (n, cout, h, w), generator=g).float()
    wi = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(xi, wi, None, 1, 1).backward(dyi)
    old = {k: os.environ.get(k) for k in ("MAAI_WGRAD_PATCH", "MAAI_WGRAD_PATCH_DB")}
    os.environ["MAAI_WGRAD_PATCH"] = "1"
    os.environ["MAAI_WGRAD_PATCH_DB"] = db
    try:
        dw = K.conv2d_wgrad(nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16), 3, 3, 1, 1, 1)
        dwi = K.conv2d_wgrad(nhwc(xi, torch.bfloat16), nhwc(dyi, torch.bfloat16), 3, 3, 1, 1, 1)
        torch.cuda.synchronize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    scale = wt.grad.abs().max().item()
    np.testing.assert_allclose(dw.cpu().permute(0, 3, 1, 2).numpy(), wt.grad.float().numpy(), rtol=1e-4, atol=2e-5 * scale + 1e-5)
    assert torch.equal(dwi.cpu().permute(0, 3, 1, 2), wi.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(4, 64, 15, 15), (2, 256, 8, 8), (3, 2048, 4, 4), (16, 1024, 1, 1)])
@pytest.mark.parametrize("res", [False, True])
def test_bn_fwd_bwd(K, shape, dtype, res):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    y = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    r = torch.randn(n, c, h, w, generator=g) if res else None
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.1
    dout = torch.randn(n, c, h, w, generator=g)
    if dtype == torch.bfloat16:
        y, dout = rb(y), rb(dout)
        r = rb(r) if res else None
    # reference (fp64 autograd, torch BatchNorm semantics)
    yd = y.double().requires_grad_(True)
    rd = r.double().requires_grad_(True) if res else None
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    o = F.batch_norm(yd, rm, rv, gd, bd, True, 0.1, 1e-5)
    o = F.relu(o + rd) if res else F.relu(o)
    o.backward(dout.double())
    # HIP: statistics from a conv-epilogue-shaped partial slab (one row here)
    M = n * h * w
    ynh = nhwc(y, dtype)
    part = torch.stack([ynh.float().sum(dim=(0, 1, 2)), (ynh.float() ** 2).sum(dim=(0, 1, 2))]).reshape(1, 2, c).contiguous()
    sums = K.reduce_partials(part)
    rmg, rvg = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    mean, invstd, scale, shift = K.bn_finalize(sums, M, gamma.cuda(), beta.cuda(), rmg, rvg, 0.1, 1e-5)
    np.testing.assert_allclose(rmg.cpu().numpy(), rm.float().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rvg.cpu().numpy(), rv.float().numpy(), rtol=1e-4, atol=1e-5)
    out = K.bn_act_fwd(ynh, scale, shift, nhwc(r, dtype) if res else None, True)
    np.testing.assert_allclose(from_nhwc(out).numpy(), o.detach().float().numpy(), **tol(dtype))
    # backward with the reference's forward output as the ReLU mask source (avoids flips at 0)
    out_ref = nhwc(o.detach().float(), dtype)
    dnh = nhwc(dout, dtype)
    s2 = K.bn_act_bwd_reduce(dnh, out_ref, ynh, mean, True)
    dgamma, dbeta, k1, k2, k3 = K.bn_bwd_coeffs(s2, M, gamma.cuda(), mean, invstd)
    dy, dz = K.bn_act_bwd_apply(dnh, out_ref, ynh, k1, k2, k3, True, True, res)
    gs = gd.grad.abs().max().item()
    np.testing.assert_allclose(dgamma.cpu().numpy(), gd.grad.float().numpy(), rtol=2e-3, atol=2e-3 * gs)
    np.testing.assert_allclose(dbeta.cpu().numpy(), bd.grad.float().numpy(), rtol=2e-3, atol=2e-3 * bd.grad.abs().max().item())
    np.testing.assert_allclose(from_nhwc(dy).numpy(), yd.grad.float().numpy(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=2e-2 if dtype == torch.bfloat16 else 1e-4)
    if res:
        np.testing.assert_allclose(from_nhwc(dz).numpy(), rd.grad.float().numpy(), **tol(dtype))


def test_bn_eval_coeffs(K):
    c = 96
    g = torch.Generator().manual_seed(3)
    gamma, beta, rm, rv = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g), torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.1
    scale, shift = K.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5)
    x = torch.randn(4, c, 3, 3, generator=g)
    ref = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    got = x * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------
# NT-Xent
# ---------------------------------------------------------------------------
def _ntxent_hip(K, h1, h2, tau, normalize=True, rank=0, world=1, gathered=None, want_dz1=False):
    h1g, h2g = h1.cuda(), h2.cuda()
    z1, inv1 = K.ntxent_normalize(h1g, normalize)
    z2, inv2 = K.ntxent_normalize(h2g, normalize)
    if world > 1:
        Z1, Z2 = gathered[0].cuda().contiguous(), gathered[1].cuda().contiguous()
    else:
        Z1, Z2 = z1, z2
    b = h1.shape[0]
    loss, logits, lse = K.ntxent_fwd(z1, z2, Z1, Z2, tau, rank * b)
    one = torch.ones((), device="cuda")
    dz1, dz2 = K.ntxent_bwd(z1, z2, Z1, Z2, lse, one, tau, rank * b, world == 1, want_dz1)
    dh2 = K.ntxent_normalize_bwd(z2, dz2, inv2, normalize)
    dh1 = K.ntxent_normalize_bwd(z1, dz1, inv1, normalize) if want_dz1 else None
    torch.cuda.synchronize()
    return loss.cpu(), logits.cpu(), dh2.cpu(), (dh1.cpu() if want_dz1 else None)


@pytest.mark.parametrize("tag", ["b8", "b16", "b64", "b33"])
def test_ntxent_golden_single(K, golden_dir, tag):
    G = np.load(os.path.join(golden_dir, "ntxent_single.npz"))
    b, d, tau, seed = G[f"{tag}_cfg"]
    b, d, seed = int(b), int(d), int(seed)
    torch.manual_seed(seed)
    h1, h2 = torch.randn(b, d), torch.randn(b, d)
    loss, logits, dh2, _ = _ntxent_hip(K, h1, h2, float(tau))
    np.testing.assert_allclose(loss.item(), G[f"{tag}_loss"], rtol=2e-6)
    np.testing.assert_allclose(logits.numpy(), G[f"{tag}_logits"], rtol=1e-5, atol=2e-6)
    # golden dh2: h1 detached (train loop), gradient through both operands of bb and the ab/ba pair
    np.testing.assert_allclose(dh2.numpy(), G[f"{tag}_dh2"], rtol=2e-4, atol=2e-7)
    # both inputs differentiable
    _, _, dh2b, dh1b = _ntxent_hip(K, h1, h2, float(tau), want_dz1=True)
    np.testing.assert_allclose(dh2b.numpy(), G[f"{tag}_dh2_both"], rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(dh1b.numpy(), G[f"{tag}_dh1_both"], rtol=2e-4, atol=2e-7)
    loss_nn, _, _, _ = _ntxent_hip(K, h1 * 0.1, h2 * 0.1, float(tau), normalize=False)
    np.testing.assert_allclose(loss_nn.item(), G[f"{tag}_loss_nonorm"], rtol=2e-6)


@pytest.mark.parametrize("world", [2, 4])
def test_ntxent_golden_multirank(K, golden_dir, world):
    G = np.load(os.path.join(golden_dir, "ntxent_gloo.npz"))
    torch.manual_seed(1234)
    H1, H2 = torch.randn(world * 8, 128), torch.randn(world * 8, 128)
    Z1, Z2 = O.l2_normalize(H1), O.l2_normalize(H2)
    for r in range(world):
        h1, h2 = H1[r * 8:(r + 1) * 8].contiguous(), H2[r * 8:(r + 1) * 8].contiguous()
        loss, logits, dh2, _ = _ntxent_hip(K, h1, h2, 0.5, True, r, world, (Z1, Z2))
        np.testing.assert_allclose(loss.item(), G[f"w{world}_loss"][r], rtol=2e-6)
        np.testing.assert_allclose(logits.numpy(), G[f"w{world}_logits"][r], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(dh2.numpy(), G[f"w{world}_dh2"][r], rtol=2e-4, atol=2e-7)


@pytest.mark.parametrize("b,n,tau", [(256, 256, 0.05), (512, 4096, 0.5), (100, 300, 0.1)])
def test_ntxent_oracle_large(K, b, n, tau):
    """BASELINE sizes (cfg2: B=N=256; cfg3: B=512, N=4096) and a ragged one against the oracle."""
    g = torch.Generator().manual_seed(b + n)
    world = n // b
    H1, H2 = torch.randn(n, 128, generator=g), torch.randn(n, 128, generator=g)
    Z1, Z2 = O.l2_normalize(H1), O.l2_normalize(H2)
    rank = world - 1
    h1, h2 = H1[rank * b:(rank + 1) * b].contiguous(), H2[rank * b:(rank + 1) * b].contiguous()
    if world == 1:
        ref_loss, ref_g = O.nt_xent_grad_h2(h1, h2, tau)
        _, ref_logits, _ = O.nt_xent(h1, h2, tau)
        loss, logits, dh2, _ = _ntxent_hip(K, h1, h2, tau)
    else:
        ref_loss, ref_g = O.nt_xent_grad_h2(h1, h2, tau, True, rank, world, (Z1, Z2))
        _, ref_logits, _ = O.nt_xent(h1, h2, tau, True, rank, world, (Z1, Z2))
        loss, logits, dh2, _ = _ntxent_hip(K, h1, h2, tau, True, rank, world, (Z1, Z2))
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=5e-6)
    np.testing.assert_allclose(logits.numpy(), ref_logits.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(dh2.numpy(), ref_g.numpy(), rtol=5e-4, atol=1e-7 + 1e-4 * ref_g.abs().max().item())


# ---------------------------------------------------------------------------
# layout, pooling, casts, optimiser, augmentation
# ---------------------------------------------------------------------------
def test_pack_views_bit_exact(K):
    views = [torch.randint(0, 256, (5, 30, 30, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(k)) for k in range(4)]
    ref = O.pack_views(views, 5, (30, 30))
    for dtype in (torch.bfloat16, torch.float32):
        out = K.pack_views_u8([v.cuda() for v in views], 32, dtype)
        got = out.float().cpu()
        assert torch.equal(got[..., :12].permute(0, 3, 1, 2), ref)
        assert torch.count_nonzero(got[..., 12:]) == 0


def test_stem_unroll_equals_7x7(K):
    """7x1 conv over the kw-unrolled operand == the reference 7x7 stride-1 pad-3 stem."""
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (2, 3, 20, 18), generator=g).float()
    w = rb(torch.randn(64, 3, 7, 7, generator=g) * 0.05)
    ref = F.conv2d(x.double(), w.double(), None, 1, 3)
    wu = torch.zeros(64, 7, 1, 32)
    for kw in range(7):
        wu[:, :, 0, kw * 4:kw * 4 + 3] = w[:, :, :, kw].permute(0, 2, 1)
    for src in (x.cuda(), x.permute(0, 2, 3, 1).contiguous().to(torch.uint8).cuda()):
        xu = K.stem_unroll(src, torch.bfloat16)
        y = K.conv2d(xu, wu.to(torch.bfloat16).cuda(), 1, 3, 0)
        np.testing.assert_allclose(from_nhwc(y).numpy(), ref.float().numpy(), rtol=1 / 128, atol=0.5)


def test_layout_roundtrip_and_pool(K):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 12, 9, 7, generator=g)
    for dtype in (torch.bfloat16, torch.float32):
        xn = K.nchw_to_nhwc(x.cuda(), 32, dtype)
        back = K.nhwc_to_nchw(xn, 12).cpu()
        assert torch.equal(back, rb(x) if dtype == torch.bfloat16 else x)
        assert torch.count_nonzero(xn[..., 12:]) == 0
    f = rb(torch.randn(2, 64, 28, 28, generator=g))
    for dtype in (torch.bfloat16, torch.float32):
        p = K.avgpool_fwd(nhwc(f, dtype), 4, 4)
        np.testing.assert_allclose(from_nhwc(p).numpy(), F.adaptive_avg_pool2d(f, (4, 4)).numpy(), **tol(dtype))
        dp = rb(torch.randn(2, 64, 4, 4, generator=g))
        fd = f.clone().requires_grad_(True)
        F.adaptive_avg_pool2d(fd, (4, 4)).backward(dp)
        dx = K.avgpool_bwd(nhwc(dp, dtype), 28, 28)
        np.testing.assert_allclose(from_nhwc(dx).numpy(), fd.grad.numpy(), rtol=1 / 128, atol=1e-4)


def test_cast_roundtrip(K):
    x = torch.randn(1000003, generator=torch.Generator().manual_seed(1))
    x[5] = float("nan")
    x[6] = float("inf")
    b = K.cast_from_f32(x.cuda(), torch.bfloat16)
    ref = x.to(torch.bfloat16)
    assert torch.equal(b.cpu().view(torch.int16)[torch.isfinite(x)], ref.view(torch.int16)[torch.isfinite(x)])
    assert torch.isnan(b.cpu()[5]) and torch.isinf(b.cpu()[6])
    assert torch.equal(K.cast_to_f32(b).cpu()[7:], ref.float()[7:])


def test_adam_matches_oracle(K):
    g = torch.Generator().manual_seed(8)
    p, gr = torch.randn(10007, generator=g), torch.randn(10007, generator=g)
    m, v = torch.zeros(10007), torch.zeros(10007)
    pg, mg, vg = p.cuda(), m.cuda(), v.cuda()
    for step in range(1, 4):
        gs = gr * step
        p, m, v = O.adam_update(p, gs, m, v, step, 1e-3)
        K.adam_step(pg, gs.cuda(), mg, vg, 1e-3, 0.9, 0.999, 1e-8, step)
    np.testing.assert_allclose(pg.cpu().numpy(), p.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vg.cpu().numpy(), v.numpy(), rtol=1e-5, atol=1e-9)
    # torch.optim.SGD with momentum + weight decay
    q = torch.nn.Parameter(torch.randn(501, generator=g))
    opt = torch.optim.SGD([q], lr=0.1, momentum=0.9, weight_decay=1e-4)
    qg, mom = q.detach().clone().cuda(), torch.zeros(501, device="cuda")
    for step in range(3):
        q.grad = torch.randn(501, generator=g)
        K.sgd_step(qg, q.grad.cuda(), mom, 0.1, 0.9, 1e-4, step == 0)
        opt.step()
    np.testing.assert_allclose(qg.cpu().numpy(), q.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_adam_multi_tensor_is_bit_identical(K):
    """HipAdam updates all tensors of a group in one launch; same bits as the per-tensor kernel and as
    torch.optim.Adam within rounding, over ragged sizes (1 element, non-multiples of the 2048-element chunk)."""
    from maai_hip.optim import HipAdam
    g = torch.Generator().manual_seed(12)
    sizes = [(1,), (7, 3), (2048,), (2049,), (64, 3, 7, 7), (5000,)]
    ps = [torch.nn.Parameter(torch.randn(sz, generator=g).cuda()) for sz in sizes]
    single = [p.detach().clone() for p in ps]
    ms, vs = [torch.zeros_like(p) for p in single], [torch.zeros_like(p) for p in single]
    ref = [torch.nn.Parameter(p.detach().clone().cpu()) for p in ps]
    opt, ropt = HipAdam(ps, lr=1e-3), torch.optim.Adam(ref, lr=1e-3)
    for step in range(1, 4):
        grads = [torch.randn(sz, generator=g) for sz in sizes]
        for p, r, gr, q, m, v in zip(ps, ref, grads, single, ms, vs):
            p.grad, r.grad = gr.cuda(), gr.clone()
            K.adam_step(q, gr.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        opt.step()
        ropt.step()
    assert len(opt._multi) == 1
    for p, q, r in zip(ps, single, ref):
        assert torch.equal(p.detach(), q)
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert opt.state[ps[-1]]["step"] == 3


def test_sgd_multi_tensor_is_bit_identical(K):
    """HipSGD (Model_Util.py:70-73) updates all tensors of a group in one launch (maai_sgd_step_multi): same bits as the
    per-tensor kernel, torch.optim.SGD(momentum, weight_decay) within rounding, ragged sizes."""
    from maai_hip.optim import HipSGD
    g = torch.Generator().manual_seed(13)
    sizes = [(1,), (7, 3), (2048,), (2049,), (64, 3, 7, 7), (5000,)]
    ps = [torch.nn.Parameter(torch.randn(sz, generator=g).cuda()) for sz in sizes]
    single = [p.detach().clone() for p in ps]
    moms = [torch.zeros_like(p) for p in single]
    ref = [torch.nn.Parameter(p.detach().clone().cpu()) for p in ps]
    opt, ropt = HipSGD(ps, lr=0.05, momentum=0.9, weight_decay=1e-4), torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        grads = [torch.randn(sz, generator=g) for sz in sizes]
        for p, r, gr, q, m in zip(ps, ref, grads, single, moms):
            p.grad, r.grad = gr.cuda(), gr.clone()
            K.sgd_step(q, gr.cuda(), m, 0.05, 0.9, 1e-4, step == 0)
        opt.step()
        ropt.step()
    assert len(opt._multi) == 1
    for p, q, r in zip(ps, single, ref):
        assert torch.equal(p.detach(), q)
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert opt.state[ps[-1]]["step"] == 3


@pytest.mark.parametrize("world,counts", [(1, [4096]), (2, [3000, 3000]), (8, [100, 7, 512, 512, 1, 9000, 33, 64])])
def test_syncbn_packed_statistics_merge(K, world, counts):
    """The fp32 SyncBatchNorm exchange (nn.SyncBatchNorm at Contrastive_Learning.py:240-252): per-rank mean | M2 | count
    rows merged with Chan's formula == BatchNorm statistics of the concatenated samples (fp64), unequal counts and a large
    mean/std ratio included; running statistics as torch updates them; non-contiguous rows (a column slice of a pair's
    exchange) accepted."""
    C = 96
    g = torch.Generator().manual_seed(21 + world)
    offs = torch.randn(C, generator=g) * 50.0       # |mean| / std up to ~150: raw sums of squares in fp32 would cancel
    xs = [torch.randn(n, C, generator=g, dtype=torch.float64) * (torch.rand(C, generator=g, dtype=torch.float64) + 0.5) + offs.double() for n in counts]
    rows = []
    for x in xs:
        sums = torch.cat([x.sum(0), (x * x).sum(0)]).cuda()
        rows.append(K.bn_pack_stats(sums, x.shape[0]))
    pad = torch.zeros(world, 5, device="cuda")
    wide = torch.cat([pad, torch.stack(rows), pad], dim=1)          # rows at a stride != 2C+1
    gathered = wide[:, 5:5 + 2 * C + 1]
    assert not gathered.is_contiguous() or world == 1
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mean, invstd, scale, shift, ncount = K.bn_finalize_gathered(gathered, gamma, beta, rm, rv, 0.1, 1e-5)
    allx = torch.cat(xs)
    n = allx.shape[0]
    assert ncount.dtype == torch.float64 and ncount.item() == float(n)   # the merged count: sum of the ranks' (unequal) counts
    m_ref, v_ref = allx.mean(0), allx.var(0, unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref.numpy(), rtol=2e-7, atol=1e-6)
    np.testing.assert_allclose(invstd.cpu().numpy(), (1.0 / torch.sqrt(v_ref + 1e-5)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(scale.cpu().numpy(), (gamma.cpu().double() / torch.sqrt(v_ref + 1e-5)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(shift.cpu().numpy(), (beta.cpu().double() - m_ref * gamma.cpu().double() / torch.sqrt(v_ref + 1e-5)).numpy(),
                               rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * m_ref.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * v_ref * n / max(n - 1, 1)).numpy(), rtol=2e-5)
    # the single-rank message reproduces maai_bn_finalize on the raw sums
    if world == 1:
        sums = torch.cat([xs[0].sum(0), (xs[0] * xs[0]).sum(0)]).cuda()
        ref = K.bn_finalize(sums, n, gamma, beta, None, None, 0.0, 1e-5)
        np.testing.assert_allclose(scale.cpu().numpy(), ref[2].cpu().numpy(), rtol=2e-5)
    # backward coefficients from fp32 sums == from fp64 sums
    s64 = torch.randn(2 * C, generator=g, dtype=torch.float64).cuda() * 100
    a = K.bn_bwd_coeffs(s64, n, gamma, mean, invstd)
    b = K.bn_bwd_coeffs(s64.float(), n, gamma, mean, invstd)
    for u, v in zip(a, b):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # ... and with the count taken from the device scalar of the merge (what the engine's SyncBatchNorm backward passes:
    # the ranks' batches may differ) exactly as with the host number
    for u, v in zip(a, K.bn_bwd_coeffs(s64, ncount, gamma, mean, invstd)):
        assert torch.equal(u, v)


def test_augment_bit_exact(K):
    g = torch.Generator().manual_seed(12)
    imgs = torch.randint(0, 256, (6, 64, 48, 3), dtype=torch.uint8, generator=g)
    params = K.augment_params(6, 64, 48, seed=77, view=1, device="cuda")
    pc = params.cpu()
    assert (pc[:, 2] <= 48).all() and (pc[:, 3] <= 64).all() and (pc[:, 0] >= 0).all() and (pc[:, 0] + pc[:, 2] <= 48 + 1e-3).all()
    assert ((pc[:, 4] == 0) | (pc[:, 4] == 1)).all()
    area = pc[:, 2] * pc[:, 3] / (64 * 48)
    assert (area > 0.05).all() and (area <= 1.0 + 1e-5).all()
    out = K.augment_view_u8(imgs.cuda(), params, 32, 32).cpu()
    for i in range(6):
        ref = O.augment_view(imgs[i], pc[i], (32, 32))
        assert torch.equal(out[i], ref), i
    # hue / saturation jitter (Contrastive_Learning.py:622-630): the colour matrix is YIQ2RGB R(hue) diag(1,s,s) RGB2YIQ
    # for SOME hue in [0, 90] degrees and saturation in [0.5, 1]; recover them from the matrix and compare with fp64
    A = torch.tensor([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]], dtype=torch.float64)
    B = torch.tensor([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]], dtype=torch.float64)
    hues = []
    for i in range(6):
        M = pc[i, 7:16].double().reshape(3, 3)
        R = torch.linalg.solve(B, M) @ torch.linalg.inv(A)      # = R(hue) diag(1, s, s)
        sat = float((R[1, 1] ** 2 + R[2, 1] ** 2).sqrt())
        hue = float(torch.atan2(R[2, 1], R[1, 1])) * 180.0 / np.pi
        assert 0.5 - 1e-3 <= sat <= 1.0 + 1e-3 and -1e-2 <= hue <= 90.0 + 1e-2, (sat, hue)
        np.testing.assert_allclose(M.numpy(), O.colour_matrix(hue, sat).numpy(), atol=2e-5)
        hues.append(hue)
    assert max(hues) - min(hues) > 5.0   # the hue really is jittered per sample
    # no jitter requested -> no hue rotation, unit saturation
    p0 = K.augment_params(4, 64, 48, seed=3, view=0, device="cuda", brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0).cpu()
    assert torch.equal(p0[:, 5], torch.ones(4)) and torch.equal(p0[:, 6], torch.ones(4))
    np.testing.assert_allclose(p0[0, 7:16].double().reshape(3, 3).numpy(), O.colour_matrix(0.0, 1.0).numpy(), atol=1e-6)
    # a pure hue rotation of 90 degrees on a saturated colour, against the fp64 matrix
    ph = p0[:1].clone()
    ph[0, :5] = torch.tensor([0.0, 0.0, 48.0, 64.0, 0.0])
    ph[0, 7:16] = O.colour_matrix(90.0, 1.0).float().reshape(-1)
    red = torch.zeros(1, 64, 48, 3, dtype=torch.uint8)
    red[..., 0] = 200
    red[..., 1] = 40
    red[..., 2] = 30
    oh = K.augment_view_u8(red.cuda(), ph.cuda(), 8, 8).cpu()[0, 0, 0].double()
    want = (O.colour_matrix(90.0, 1.0) @ torch.tensor([200.0, 40.0, 30.0], dtype=torch.float64)).clamp(0, 255)
    assert (oh - want).abs().max() <= 1.0, (oh, want)
    # different view id -> different parameters, same seed+view -> identical (counter-based)
    p2 = K.augment_params(6, 64, 48, seed=77, view=2, device="cuda").cpu()
    p1 = K.augment_params(6, 64, 48, seed=77, view=1, device="cuda").cpu()
    assert torch.equal(p1, pc) and not torch.equal(p2, pc)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_conv_relu_mask_epilogue(K, dtype, conv_bm, conv_halo):
    """y = conv(x, w) * (mask > 0), alone and combined with accumulate + strided scatter."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 64, 9, 9, generator=g)
    wt = torch.randn(128, 64, 3, 3, generator=g) / 24
    mask = torch.randn(2, 128, 9, 9, generator=g)
    if dtype == torch.bfloat16:
        x, wt, mask = rb(x), rb(wt), rb(mask)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1) * (mask > 0)
    y = K.conv2d(nhwc(x, dtype), khwc(wt, dtype), 1, 1, 1, relu_mask=nhwc(mask, dtype))
    np.testing.assert_allclose(from_nhwc(y).numpy(), ref.float().numpy(), **tol(dtype))
    base = torch.randn(2, 128, 17, 17, generator=g)
    mask2 = torch.randn(2, 128, 17, 17, generator=g)
    if dtype == torch.bfloat16:
        base, mask2 = rb(base), rb(mask2)
    out = nhwc(base, dtype)
    K.conv2d(nhwc(x, dtype), khwc(wt, dtype), 1, 1, 1, out=out, grid_hw=(9, 9), out_hw=(17, 17), out_stride=2, accumulate=True,
             relu_mask=nhwc(mask2, dtype))
    ref2 = base.clone().double()
    ref2[:, :, 0::2, 0::2] = (ref2[:, :, 0::2, 0::2] + F.conv2d(x.double(), wt.double(), None, 1, 1)) * (mask2[:, :, 0::2, 0::2] > 0)
    np.testing.assert_allclose(from_nhwc(out).numpy(), ref2.float().numpy(), **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("variant", ["mask_tensor", "mask_from_y", "mask_bits", "no_mask"])
@pytest.mark.parametrize("case", [(2, 64, 9, 9, 128, 3, False), (3, 256, 7, 5, 64, 1, True), (2, 128, 8, 8, 64, 3, True), (1, 64, 40, 40, 192, 3, False)])
def test_conv_store_reduce_epilogue(K, case, variant, dtype, conv_bm, conv_halo):
    """MAAI_EPI_DGRAD_REDUCE: the stored gradient g equals the plain (accumulate + mask) epilogue's bit for bit,
    and the partial sums equal those of the separate reduction pass over the stored g (sum g, sum g*(y - mean))."""
    n, cin, h, w, cout, k, acc = case
    g = torch.Generator().manual_seed(31 + cin + k)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ylow = torch.randn(n, cout, h, w, generator=g)
    base = torch.randn(n, cout, h, w, generator=g)
    mean, scale, shift = torch.randn(cout, generator=g), torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.3
    xd, wd, yd, bd = nhwc(x, dtype), khwc(wt, dtype), nhwc(ylow, dtype), nhwc(base, dtype)
    mean_d, scale_d, shift_d = mean.cuda(), scale.cuda(), shift.cuda()
    p = k // 2
    bits = None
    if variant == "mask_bits":
        if dtype != torch.bfloat16:
            pytest.skip("the 1-bit mask exists for bf16 only")
        res = nhwc(torch.randn(n, cout, h, w, generator=g), dtype)
        mask, bits = K.bn_act_fwd(yd, scale_d, shift_d, res, True, want_bits=True)   # a residual unit's output + its bits
        assert torch.equal(mask, K.bn_act_fwd(yd, scale_d, shift_d, res, True))
        packed = np.packbits((mask.float().cpu().numpy().reshape(-1) > 0).astype(np.uint8), bitorder="little")
        assert np.array_equal(bits.cpu().numpy(), packed)
        sc = sh = None
    elif variant == "mask_tensor":
        mask = nhwc(torch.randn(n, cout, h, w, generator=g), dtype)
        sc = sh = None
    elif variant == "mask_from_y":
        mask = K.bn_act_fwd(yd, scale_d, shift_d, None, True)      # the ReLU output the mask stands for
        sc, sh = scale_d, shift_d
    else:
        mask = sc = sh = None
    want = bd.clone() if acc else torch.empty_like(bd)
    K.conv2d(xd, wd, 1, p, p, out=want, grid_hw=(h, w), out_hw=(h, w), accumulate=acc, relu_mask=mask)
    got = bd.clone() if acc else torch.empty_like(bd)
    rows = K.conv2d_stats_rows(xd, wd, 1, p, p, (h, w), (h, w))
    part = torch.full((rows, 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    K.conv2d_store_reduce(xd, wd, 1, p, p, got, part, yd, mean_d, sc, sh,
                          None if variant == "mask_from_y" else (bits if bits is not None else mask),
                          grid_hw=(h, w), out_hw=(h, w), accumulate=acc, mask_bits=bits is not None)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    sums = K.reduce_partials(part).cpu()
    wf, yf = want.double().reshape(-1, cout), yd.double().reshape(-1, cout)
    ref = torch.cat([wf.sum(0), (wf * (yf - mean_d.double())).sum(0)]).cpu()
    np.testing.assert_allclose(sums.numpy(), ref.numpy(), rtol=1e-4, atol=1e-3 if dtype == torch.float32 else 2e-2)
    if (cout // (8 if dtype == torch.bfloat16 else 4)) in (8, 16, 32, 64, 128, 256):
        sep = K.bn_act_bwd_reduce(want, None, yd, mean_d, False).cpu()
        np.testing.assert_allclose(sums.numpy(), sep.numpy(), rtol=2e-5, atol=2e-4)


@pytest.fixture(params=["0", "2"], ids=["ring", "streaming"])
def pw_direct(request):
    """Both pointwise kernels: the ring kernel (conv_igemm.h) and, wherever it is built for the shape, the streaming
    kernel (conv_pws.hip; MAAI_CONV_PWS=2 lifts its shape rule)."""
    old = os.environ.get("MAAI_CONV_PWS")
    os.environ["MAAI_CONV_PWS"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("MAAI_CONV_PWS", None)
    else:
        os.environ["MAAI_CONV_PWS"] = old


@pytest.mark.parametrize("case", [(2, 64, 30, 30, 256), (3, 256, 15, 15, 64), (1, 512, 4, 4, 2048), (5, 96, 9, 7, 192)])
def test_pointwise_kernels_agree(K, case, pw_direct):
    """1x1 layers through either kernel: forward + statistics, and accumulate + ReLU-mask store."""
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    base = rb(torch.randn(n, cout, h, w, generator=g))
    mask = rb(torch.randn(n, cout, h, w, generator=g))
    ref = F.conv2d(x.double(), wt.double())
    y, part = K.conv2d(nhwc(x, torch.bfloat16), khwc(wt, torch.bfloat16), 1, 0, 0, stats=True)
    np.testing.assert_allclose(from_nhwc(y).numpy(), ref.float().numpy(), **tol(torch.bfloat16))
    sums = K.reduce_partials(part).cpu()
    np.testing.assert_allclose(sums[:cout].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=2e-3 * (n * h * w) ** 0.5)
    np.testing.assert_allclose(sums[cout:].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=2e-4, atol=1e-3)
    out = nhwc(base, torch.bfloat16)
    K.conv2d(nhwc(x, torch.bfloat16), khwc(wt, torch.bfloat16), 1, 0, 0, out=out, accumulate=True, relu_mask=nhwc(mask, torch.bfloat16))
    ref2 = (rb(ref.float()).double() + base.double()) * (mask > 0)
    np.testing.assert_allclose(from_nhwc(out).numpy(), ref2.float().numpy(), **tol(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", [(3, 64, 15, 15, 256), (2, 128, 9, 7, 512), (1, 32, 5, 5, 64)])
def test_conv_fused_bn_epilogues(K, case, dtype, pw_direct):
    """The fused pointwise unit (stats-only pass, BN+residual+ReLU epilogue, BN-backward reduce / apply with
    the convolution recomputed) against the unfused kernels it replaces — same roundings, so the forward must
    agree to 1 ulp and the backward sums to fp32 re-association."""
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    res = torch.randn(n, cout, h, w, generator=g)
    dz = torch.randn(n, cout, h, w, generator=g)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    mean = torch.randn(cout, generator=g) * 0.1
    k1, k2, k3 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x, wt, res, dz = rb(x), rb(wt), rb(res), rb(dz)
    xg, wg, rg, dzg = nhwc(x, dtype), khwc(wt, dtype), nhwc(res, dtype), nhwc(dz, dtype)
    sc, sh, mu, k1g, k2g, k3g = (t.cuda() for t in (scale, shift, mean, k1, k2, k3))
    # unfused references built from the already-verified kernels
    y, part = K.conv2d(xg, wg, 1, 0, 0, stats=True)
    part2 = K.conv2d_stats_only(xg, wg)
    if pw_direct == "0":
        assert torch.equal(part, part2)
    else:   # the streaming kernel sums the rows of a tile in another order
        assert part.shape == part2.shape
        np.testing.assert_allclose(part.sum(0).cpu().numpy(), part2.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
    out_ref = K.bn_act_fwd(y, sc, sh, rg, True)
    out = K.conv2d_bn_act(xg, wg, sc, sh, rg, True)
    assert torch.equal(out, out_ref)
    out_nr = K.conv2d_bn_act(xg, wg, sc, sh, None, False)
    assert torch.equal(out_nr, K.bn_act_fwd(y, sc, sh, None, False))
    sums_ref = K.bn_act_bwd_reduce(dzg, None, y, mu, False).cpu()
    sums = K.conv2d_bwd_reduce(xg, wg, dzg, mu).cpu()
    np.testing.assert_allclose(sums.numpy(), sums_ref.numpy(), rtol=1e-4, atol=1e-3)
    dy_ref, _ = K.bn_act_bwd_apply(dzg, None, y, k1g, k2g, k3g, False, True, False)
    dy = K.conv2d_bwd_apply(xg, wg, dzg, k1g, k2g, k3g)
    assert torch.equal(dy, dy_ref)


def test_foveated_retinal_processor_matches_oracle(K):
    """One kernel for the whole DALI fixation graph vs its numpy restatement: every u8 within 1 LSB."""
    from maai_hip import foveated
    g = torch.Generator().manual_seed(33)
    B, H, W = 5, 96, 128
    imgs = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    hw = np.array([[96, 128], [80, 100], [96, 64], [50, 128], [96, 128]])
    rng = np.random.default_rng(15)
    P = foveated.build_params(B, hw, rng, pos_x=torch.rand(B, 1, generator=g), pos_y=torch.rand(B, 1, generator=g),
                              angle=(torch.rand(B, 1, generator=g) - 0.5) * 160, gm_ratio=torch.tensor([0.3, 0.0, 0.45, 0.2, 0.0]),
                              gm_tile=torch.tensor([120, 1, 300, 450, 1]), noise_mean=torch.tensor([0.2, 0.0, -0.3, 0.0, 0.1]),
                              noise_std=torch.tensor([30.0, 0.0, 80.0, 5.0, 0.0]),
                              brightness=0.6 + 0.8 * torch.rand(B, 1, generator=g), contrast=0.6 + 0.8 * torch.rand(B, 1, generator=g),
                              hue=torch.rand(B, 1, generator=g) * 0.5 * 360, saturation=0.2 + 0.8 * torch.rand(B, 1, generator=g))
    views = foveated.foveate(imgs.cuda(), P)
    ref = O.foveate_views(imgs.numpy(), P.numpy())
    assert len(views) == 4
    for v, r in zip(views, ref):
        got = v.cpu().numpy().astype(np.int32)
        assert got.shape == (B, 30, 30, 3)
        d = np.abs(got - r.astype(np.int32))
        assert d.max() <= 1, d.max()
        assert (d > 0).mean() < 0.01
    # labelled (evaluation) variant: centre crop, no flip, identity colour -> the 30-px view is a plain window
    P2 = foveated.build_params(B, hw, rng, pos_x=torch.full((B, 1), 0.5), pos_y=torch.full((B, 1), 0.5), labeled=True)
    v2 = foveated.foveate(imgs.cuda(), P2)
    r2 = O.foveate_views(imgs.numpy(), P2.numpy())
    for v, r in zip(v2, r2):
        assert np.abs(v.cpu().numpy().astype(np.int32) - r.astype(np.int32)).max() <= 1
    assert not np.array_equal(views[3].cpu().numpy(), v2[3].cpu().numpy())


def test_batched_weight_forms_match_the_torch_construction(K):
    """engine.w_fwd / w_dgrad on device parameters go through ONE-launch conversion (maai_weight_forms): same bits
    as permute + index + cast, for full and subset tap lists, channel padding, fp32 and bf16; the whole registry is
    refreshed (in place) when any parameter changes."""
    from maai_hip import engine
    engine.clear_weight_cache()
    g = torch.Generator().manual_seed(4)
    w3 = torch.nn.Parameter(torch.randn(128, 64, 3, 3, generator=g).cuda())
    w1 = torch.nn.Parameter(torch.randn(256, 64, 1, 1, generator=g).cuda())
    w12 = torch.nn.Parameter(torch.randn(64, 12, 7, 7, generator=g).cuda())

    def ref_fwd(w, dtype, pad=None):
        t = w.detach().permute(0, 2, 3, 1)
        if pad:
            t = torch.nn.functional.pad(t, (0, pad - t.shape[3]))
        return t.contiguous().to(dtype)

    def ref_dgrad(w, dtype, khs, kws):
        return w.detach()[:, :, khs][:, :, :, kws].permute(1, 2, 3, 0).contiguous().to(dtype)
    for dtype in (torch.bfloat16, torch.float32):
        assert torch.equal(engine.w_fwd(w3, dtype), ref_fwd(w3, dtype))
        assert torch.equal(engine.w_fwd(w1, dtype), ref_fwd(w1, dtype))
        assert torch.equal(engine.w_fwd(w12, dtype, 32), ref_fwd(w12, dtype, 32))
        assert torch.equal(engine.w_dgrad(w3, dtype, [2, 1, 0], [2, 1, 0]), ref_dgrad(w3, dtype, [2, 1, 0], [2, 1, 0]))
        assert torch.equal(engine.w_dgrad(w3, dtype, [1], [2, 0]), ref_dgrad(w3, dtype, [1], [2, 0]))
        assert torch.equal(engine.w_dgrad(w1, dtype, [0], [0]), ref_dgrad(w1, dtype, [0], [0]))
    a = engine.w_fwd(w3, torch.bfloat16)
    b = engine.w_dgrad(w1, torch.bfloat16, [0], [0])
    with torch.no_grad():
        w3.mul_(2.0)
        w1.add_(1.0)
    assert engine.w_fwd(w3, torch.bfloat16) is a and torch.equal(a, ref_fwd(w3, torch.bfloat16))
    assert torch.equal(b, ref_dgrad(w1, torch.bfloat16, [0], [0]))   # refreshed by the same launch
    del w12
    w5 = torch.nn.Parameter(torch.randn(64, 32, 3, 3, generator=g).cuda())
    assert torch.equal(engine.w_fwd(w5, torch.bfloat16), ref_fwd(w5, torch.bfloat16))   # registration purges dead entries
    assert len(engine._FORMS["reg"].entries) == len(engine._FORMS["index"])
    engine.clear_weight_cache()


@pytest.mark.parametrize("case", [(2, 256, 9, 9, 64, False), (3, 64, 7, 5, 256, True), (1, 512, 12, 12, 128, False), (2, 128, 30, 30, 192, True)])
def test_conv_transformed_operand_matches_apply_then_conv(K, case):
    """DGRAD_REDUCE with the BatchNorm-backward apply formed while staging the A operand (k1*dz - k2 - k3*y):
    the stored gradient, the handed-back operand dy and the sums against the two-launch sequence
    maai_bn_act_bwd_apply -> plain DGRAD_REDUCE launch.  Same arithmetic, so dx and dy must be bit-identical."""
    n, cin, h, w, cout, acc = case
    g = torch.Generator().manual_seed(91 + cin)
    dt = torch.bfloat16
    dz = nhwc(torch.randn(n, cin, h, w, generator=g), dt)
    yup = nhwc(torch.randn(n, cin, h, w, generator=g), dt)
    k1, k2, k3 = (torch.randn(cin, generator=g).cuda() for _ in range(3))
    wq = khwc(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    ylow = nhwc(torch.randn(n, cout, h, w, generator=g), dt)
    base = nhwc(torch.randn(n, cout, h, w, generator=g), dt)
    mean, scale, shift = torch.randn(cout, generator=g).cuda(), torch.randn(cout, generator=g).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    # reference: apply pass, then the plain fused launch
    dy_ref, _ = K.bn_act_bwd_apply(dz, None, yup, k1, k2, k3, False, True, False)
    want = base.clone() if acc else torch.empty_like(base)
    rows = K.conv2d_stats_rows(dy_ref, wq, 1, 0, 0, (h, w), (h, w))
    part_ref = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
    K.conv2d_store_reduce(dy_ref, wq, 1, 0, 0, want, part_ref, ylow, mean, scale, shift, None, grid_hw=(h, w), out_hw=(h, w), accumulate=acc)
    # fused
    got = base.clone() if acc else torch.empty_like(base)
    dy = torch.full_like(dz, float("nan"))
    rows2 = K.conv2d_stats_rows(dz, wq, 1, 0, 0, (h, w), (h, w), axf=True)
    part = torch.full((rows2, 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    K.conv2d_store_reduce(dz, wq, 1, 0, 0, got, part, ylow, mean, scale, shift, None, grid_hw=(h, w), out_hw=(h, w), accumulate=acc,
                          axf=(yup, k1, k2, k3, dy))
    torch.cuda.synchronize()
    assert torch.equal(dy, dy_ref)
    assert torch.equal(got, want)
    np.testing.assert_allclose(K.reduce_partials(part).cpu().numpy(), K.reduce_partials(part_ref).cpu().numpy(), rtol=2e-5, atol=2e-4)
    # without the side output
    got2 = base.clone() if acc else torch.empty_like(base)
    K.conv2d_store_reduce(dz, wq, 1, 0, 0, got2, part, ylow, mean, scale, shift, None, grid_hw=(h, w), out_hw=(h, w), accumulate=acc,
                          axf=(yup, k1, k2, k3, None))
    assert torch.equal(got2, want)


@pytest.mark.parametrize("case", [(2, 64, 30, 30, 256, 1), (1, 512, 4, 4, 2048, 1), (2, 256, 15, 15, 512, 2), (3, 128, 17, 13, 512, 1)])
def test_wide_column_tile_is_bit_identical(K, case):
    """128x256 tiles (chosen by shape for the channel-expanding 1x1 layers; forced here with MAAI_CONV_BN=256) against
    128x128 tiles: outputs, BatchNorm partial statistics, and the accumulate + mask + BN-backward-sums epilogue."""
    n, cin, h, w, cout, stride = case
    g = torch.Generator().manual_seed(23 + cin)
    dt = torch.bfloat16
    x = nhwc(torch.randn(n, cin, h, w, generator=g), dt)
    wq = khwc(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    ylow = nhwc(torch.randn(n, cout, oh, ow, generator=g), dt)
    base = nhwc(torch.randn(n, cout, oh, ow, generator=g), dt)
    mean, scale, shift = torch.randn(cout, generator=g).cuda(), torch.randn(cout, generator=g).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    old = os.environ.get("MAAI_CONV_BN")
    got = {}
    try:
        for bn in ("128", "256"):
            os.environ["MAAI_CONV_BN"] = bn
            y, part = K.conv2d(x, wq, stride, 0, 0, stats=True)
            res = [y.clone(), part.clone()]
            if stride == 1:
                out = base.clone()
                rows = K.conv2d_stats_rows(x, wq, 1, 0, 0, (oh, ow), (oh, ow))
                p2 = torch.zeros((rows, 2, cout), dtype=torch.float32, device="cuda")
                K.conv2d_store_reduce(x, wq, 1, 0, 0, out, p2, ylow, mean, scale, shift, None, grid_hw=(oh, ow), out_hw=(oh, ow), accumulate=True)
                res += [out, K.reduce_partials(p2)]
            torch.cuda.synchronize()
            got[bn] = res
    finally:
        if old is None:
            os.environ.pop("MAAI_CONV_BN", None)
        else:
            os.environ["MAAI_CONV_BN"] = old
    assert torch.equal(got["128"][0], got["256"][0])
    assert torch.equal(got["128"][1], got["256"][1])
    if stride == 1:
        assert torch.equal(got["128"][2], got["256"][2])
        np.testing.assert_allclose(got["256"][3].cpu().numpy(), got["128"][3].cpu().numpy(), rtol=2e-5, atol=2e-4)
    ref = F.conv2d(from_nhwc(x).double(), wq.float().cpu().permute(0, 3, 1, 2).double(), None, stride)
    np.testing.assert_allclose(from_nhwc(got["256"][0]).numpy(), ref.float().numpy(), **tol(dt))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_store_reduce_sum_increment_completes_the_sums(K, dtype):
    """Stride-2 shortcut gradient: a dense pass reduces the sums of what it stores, a strided accumulate pass with
    sum_increment those of what it adds; together they must equal the sums of the final tensor (exactly, up to the
    fp32 order of partial sums), and the tensor must equal the plain two-pass result bit for bit."""
    g = torch.Generator().manual_seed(77)
    n, c, h, w = 2, 128, 13, 11
    h2, w2 = (h + 1) // 2, (w + 1) // 2
    xa = nhwc(torch.randn(n, 64, h, w, generator=g), dtype)
    wa = khwc(torch.randn(c, 64, 1, 1, generator=g) / 8, dtype)
    xb = nhwc(torch.randn(n, 256, h2, w2, generator=g), dtype)
    wb = khwc(torch.randn(c, 256, 1, 1, generator=g) / 16, dtype)
    ylow = nhwc(torch.randn(n, c, h, w, generator=g), dtype)
    mask = nhwc(torch.randn(n, c, h, w, generator=g), dtype)
    mean = torch.randn(c, generator=g).cuda()
    # plain: dense masked pass, then strided accumulate + mask
    want = torch.empty((n, h, w, c), dtype=dtype, device="cuda")
    K.conv2d(xa, wa, 1, 0, 0, out=want, relu_mask=mask)
    K.conv2d(xb, wb, 1, 0, 0, out=want, grid_hw=(h2, w2), out_hw=(h, w), out_stride=2, accumulate=True, relu_mask=mask)
    # fused: both passes reduce
    got = torch.empty_like(want)
    r1 = K.conv2d_stats_rows(xa, wa, 1, 0, 0, (h, w), (h, w))
    p1 = torch.zeros((r1, 2, c), dtype=torch.float32, device="cuda")
    K.conv2d_store_reduce(xa, wa, 1, 0, 0, got, p1, ylow, mean, None, None, mask, grid_hw=(h, w), out_hw=(h, w))
    r2 = K.conv2d_stats_rows(xb, wb, 1, 0, 0, (h2, w2), (h, w), 2)
    p2 = torch.zeros((r2, 2, c), dtype=torch.float32, device="cuda")
    K.conv2d_store_reduce(xb, wb, 1, 0, 0, got, p2, ylow, mean, None, None, mask, grid_hw=(h2, w2), out_hw=(h, w), out_stride=2,
                          accumulate=True, sum_increment=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    sums = (K.reduce_partials(p1) + K.reduce_partials(p2)).cpu()
    wf, yf = want.double().reshape(-1, c), ylow.double().reshape(-1, c)
    ref = torch.cat([wf.sum(0), (wf * (yf - mean.double())).sum(0)]).cpu()
    np.testing.assert_allclose(sums.numpy(), ref.numpy(), rtol=1e-4, atol=2e-3 if dtype == torch.float32 else 3e-2)


def test_softmax_ce_kernel(K):
    """maai_softmax_ce_fwd / _bwd against torch (nn.CrossEntropyLoss semantics: mean reduction, class-index targets),
    including padded logit columns (ld > C) and extreme logits."""
    g = torch.Generator().manual_seed(3)
    for (b, c, ld) in [(8, 1000, 1024), (5, 10, 64), (33, 64, 64), (256, 1000, 1024)]:
        logits = torch.randn(b, ld, generator=g) * 3
        logits[0, :c] += 40.0                       # large values: the kernel subtracts the row maximum
        logits[:, c:] = 1e6                          # padding columns must be ignored
        labels = torch.randint(0, c, (b,), generator=g)
        lt = logits[:, :c].clone().double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(lt, labels)
        ref.backward()
        loss, lse = K.softmax_ce_fwd(logits.cuda(), labels.cuda(), c)
        d = K.softmax_ce_bwd(logits.cuda(), labels.cuda(), lse, torch.tensor([2.0]).cuda(), c)
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
        np.testing.assert_allclose(d[:, :c].cpu().numpy(), 2.0 * lt.grad.float().numpy(), rtol=2e-5, atol=1e-8)
        assert float(d[:, c:].abs().max()) == 0.0 if ld > c else True


def test_probe_linear_against_torch(K):
    from maai_hip import probe
    g = torch.Generator().manual_seed(4)
    for (b, i, o) in [(8, 512, 1000), (16, 8192, 10), (3, 64, 64)]:
        x = torch.randn(b, i, generator=g)
        w = torch.randn(o, i, generator=g) / i ** 0.5
        bias = torch.randn(o, generator=g)
        dy = torch.randn(b, o, generator=g)
        xt, wt, bt = (t.clone().double().requires_grad_(True) for t in (x, w, bias))
        (torch.nn.functional.linear(xt, wt, bt) * dy.double()).sum().backward()
        xg, wg, bg = (t.clone().cuda().requires_grad_(True) for t in (x, w, bias))
        y = probe.linear(xg, wg, bg)
        assert y.shape == (b, o)
        (y * dy.cuda()).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), torch.nn.functional.linear(x, w, bias).numpy(), rtol=2e-5, atol=2e-5)
        for got, ref in ((xg.grad, xt.grad), (wg.grad, wt.grad), (bg.grad, bt.grad)):
            np.testing.assert_allclose(got.cpu().numpy(), ref.float().numpy(), rtol=2e-4, atol=2e-5)


def test_larc_multi_tensor_kernels(K):
    """maai_multi_sqnorm / maai_larc_scale (one launch each for all tensors) against the per-tensor formula of
    apex.parallel.LARC as published (trust 0.02, clip); LARC parity itself stays unpinned (Apex not installable)."""
    g = torch.Generator().manual_seed(6)
    shapes = [(64, 3, 7, 7), (64,), (5000, 13), (1,), (2048, 3), (300,)]
    ps = [torch.randn(s, generator=g).cuda() * (0.1 + i) for i, s in enumerate(shapes)]
    gs = [torch.randn(s, generator=g).cuda() * 0.01 for s in shapes]
    gs[3].zero_()                                    # a zero gradient norm: that tensor is left alone
    multi = K.LarcMulti(ps)
    norms = multi.sqnorms(gs).clone()
    for i, (p, gr) in enumerate(zip(ps, gs)):
        np.testing.assert_allclose(norms[i, 0].item(), float((p.double() ** 2).sum()), rtol=1e-5)
        np.testing.assert_allclose(norms[i, 1].item(), float((gr.double() ** 2).sum()), rtol=1e-5, atol=1e-30)
    trust, lr, wd, eps = 0.02, 0.1, 1e-4, 1e-8
    want = []
    for p, gr in zip(ps, gs):
        pn, gn = p.norm().item(), gr.norm().item()
        if pn != 0 and gn != 0:
            rate = min(trust * pn / (gn + pn * wd + eps) / lr, 1.0)
            want.append((gr + wd * p) * rate)
        else:
            want.append(gr.clone())
    multi.scale(trust, lr, wd, eps, True)
    for got, ref in zip(gs, want):
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=1e-9)
