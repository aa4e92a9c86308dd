"""BatchNorm-backward folded through a channel-expanding pointwise convolution (engine._FOLD, csrc/fold.hip): the backward of
conv3 + bn3 of the reference's Bottleneck (SimCLR/ResNet/resnet.py:118-119, training-mode nn.BatchNorm2d) WITHOUT the
BatchNorm-backward apply pass and without reading the raw convolution output y —

    G1 = g^T x,  S2 = rowsum(W * G1) - mean * S1,  dW = k1*G1 - k2 (x) colsum(x) - k3*(W Gram),  dx = g (k1 W) - x (W^T diag(k3) W) - k2 W

The folded path is NOT bit-identical to the unfolded one (y is never rounded to bf16; the folded weights are): it is pinned
here against an fp64 torch-autograd evaluation of the same unit on the same bf16 inputs, with the unfolded HIP path beside
it as the yardstick: the folded gradients must be at least as close to fp64 as 1.5 x the unfolded path's error + 2^-9 of
the gradient's scale (the bf16 storage rounding of the result itself)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from maai_hip import engine
    return engine


@pytest.fixture(autouse=True)
def _restore(E):
    yield
    E.set_fold(True)
    E.set_precision("bf16")


@pytest.mark.parametrize("cout,cin", [(256, 64), (512, 128), (2048, 512), (512, 256)])
def test_fold_algebra_kernels_match_fp64(E, cout, cin):
    """maai_fold_s2 / maai_fold_dw / maai_fold_dgrad_w against their definitions in fp64."""
    from maai_hip import kernels as K
    g = torch.Generator().manual_seed(cout + cin)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda().bfloat16()
    g1 = torch.randn(cout, cin, generator=g).cuda() * 30
    gram = torch.randn(cin, cin, generator=g).cuda() * 100
    gram = gram + gram.t()
    sx = (torch.randn(2 * cin, generator=g, dtype=torch.float64) * 50).cuda()
    s1 = (torch.randn(cout, generator=g, dtype=torch.float64) * 10).cuda()
    mean = torch.randn(cout, generator=g).cuda()
    k1, k2, k3 = (torch.randn(cout, generator=g).cuda() * s for s in (1.0, 1e-3, 1e-4))
    wd = w.double()
    s2 = K.fold_s2(w, g1, s1, mean)
    ref = (wd * g1.double()).sum(1) - mean.double() * s1
    assert (s2 - ref).abs().max() <= 1e-9 * ref.abs().max()
    dw = K.fold_dw(w, g1, gram, sx, k1, k2, k3)
    ref = k1.double()[:, None] * g1.double() - k2.double()[:, None] * sx[:cin][None, :] - k3.double()[:, None] * (wd @ gram.double())
    assert (dw.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    npix = 5000.0
    wf, tn, cn = K.fold_dgrad_weights(w, k1, k2, k3, s1, sx, npix)
    assert tuple(wf.shape) == (cin, 1, 1, cout) and tuple(tn.shape) == (cin, 1, 1, cin)
    ref_wf = (k1[:, None] * w.float()).t().contiguous()
    assert torch.equal(wf.reshape(cin, cout), ref_wf.bfloat16())
    ref_t = -(wd.t() @ (k3.double()[:, None] * wd))
    assert (tn.reshape(cin, cin).double() - ref_t.t()).abs().max() <= 2.0 ** -8 * ref_t.abs().max()   # bf16 result: half an ulp = 2^-9
    # the constant: -(k2 W) minus the pixel mean of what the two roundings add to dx[:, k]
    comp = ((wf.reshape(cin, cout).double() - ref_wf.double()) @ s1 + (tn.reshape(cin, cin).double() - ref_t.t()) @ sx[:cin]) / npix
    ref = -(k2.double()[None, :] @ wd).reshape(cin) - comp
    assert (cn.double() - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-9


def _unit(E, n, h, w_, cin, cout, seed):
    """One conv3-like unit with a conv2-like unit below it, on random bf16 tensors: returns everything the engine's unit_bwd
    needs and the fp64 reference gradients."""
    from maai_hip import kernels as K
    g = torch.Generator().manual_seed(seed)
    conv2 = torch.nn.Conv2d(cin, cin, 3, padding=1, bias=False).cuda()
    bn2 = torch.nn.BatchNorm2d(cin).cuda().train()
    conv3 = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    bn3 = torch.nn.BatchNorm2d(cout).cuda().train()
    with torch.no_grad():
        conv3.weight.copy_((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda())
        bn3.weight.copy_((torch.rand(cout, generator=g) + 0.5).cuda())
        bn3.bias.copy_((torch.randn(cout, generator=g) * 0.3).cuda())
        bn2.weight.copy_((torch.rand(cin, generator=g) + 0.5).cuda())
        bn2.bias.copy_((torch.randn(cin, generator=g) * 0.3).cuda())
    x1 = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()      # conv2's input (a post-ReLU tensor in the net)
    x1 = torch.relu(x1)
    dtype = torch.bfloat16
    a2, r2 = E.unit_fwd(x1, conv2, bn2, True, None, dtype, True)
    out, r3 = E.unit_fwd(a2, conv3, bn3, True, None, dtype, True)
    gout = torch.randn(out.shape, generator=g).cuda().bfloat16()
    gout = gout + 0.3                                                   # a gradient with a mean: k2 matters
    return dict(conv2=conv2, bn2=bn2, conv3=conv3, bn3=bn3, x1=x1, a2=a2, r2=r2, r3=r3, out=out, gout=gout)


def _reference(u):
    """fp64 autograd through conv3 -> bn3 (training) on the HIP path's own bf16 a2 and weights; returns gradients wrt a2 (masked
    by a2 > 0, the engine's convention), conv3.weight, bn3.weight, bn3.bias and bn2's backward sums of the masked gradient."""
    a2 = u["a2"].double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w = u["conv3"].weight.detach().bfloat16().double().requires_grad_(True)
    gam = u["bn3"].weight.detach().double().requires_grad_(True)
    bet = u["bn3"].bias.detach().double().requires_grad_(True)
    y = torch.nn.functional.conv2d(a2, w)
    z = torch.nn.functional.batch_norm(y, None, None, gam, bet, True, 0.1, 1e-5)
    gz = (u["gout"].double() * (u["out"].double() > 0)).permute(0, 3, 1, 2)
    z.backward(gz)
    dx = (a2.grad * (a2.detach() > 0)).permute(0, 2, 3, 1).contiguous()
    y2 = u["r2"].y.double()
    mean2 = u["r2"].mean.double()
    sums = torch.cat([dx.sum((0, 1, 2)), (dx * (y2 - mean2)).sum((0, 1, 2))])
    return dx, w.grad, gam.grad, bet.grad, sums


def _run(E, u, fold):
    E.set_fold(fold)
    grads = {}
    gm = E.relu_mask_grad(u["gout"].clone(), u["out"])
    dx, sums = E.unit_bwd(u["r3"], gm, grads, torch.bfloat16, below=u["r2"])
    torch.cuda.synchronize()
    return dx, grads[id(u["conv3"].weight)], grads[id(u["bn3"].weight)], grads[id(u["bn3"].bias)], sums


@pytest.mark.parametrize("shape", [(4, 28, 28, 128, 512), (8, 14, 14, 256, 1024), (8, 7, 7, 512, 2048), (2, 33, 35, 64, 256)],
                         ids=lambda s: "x".join(map(str, s)))
def test_folded_unit_backward_against_fp64(E, shape):
    n, h, w_, cin, cout = shape
    u = _unit(E, n, h, w_, cin, cout, seed=sum(shape))
    assert not isinstance(u["r3"].x, type(None))
    ref = _reference(u)
    assert E._fold_applies(u["r3"], u["gout"])
    got_f = _run(E, u, True)
    assert u["r3"].fold is None
    got_u = _run(E, u, False)
    names = ("dx", "dW", "dgamma", "dbeta", "sums below")
    for nm, f, un, r in zip(names, got_f, got_u, ref):
        f, un, r = f.double().reshape(-1), un.double().reshape(-1), r.reshape(-1).to(f.device)
        scale = r.abs().max().item()
        ef, eu = (f - r).abs().max().item() / scale, (un - r).abs().max().item() / scale
        cos = torch.nn.functional.cosine_similarity(f, r, dim=0).item()
        print("%s %s: folded %.3e  unfolded %.3e  cos %.6f" % ("x".join(map(str, shape)), nm, ef, eu, cos))
        assert cos > 0.9999, (nm, cos)
        assert ef <= 1.5 * eu + 2.0 ** -8, (nm, ef, eu)


def test_folded_unit_backward_in_a_projection_block(E):
    """A whole Bottleneck with a projection shortcut (the dual-BatchNorm path of block_bwd: the folded main branch, the
    strided shortcut on the ordinary apply pass) and one with an identity shortcut, folded vs unfolded gradients of every
    parameter and of the block input: same direction (cos > 0.9995) and scale (1 %)."""
    import sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in ("multimodal-active-ai_amd/SimCLR/ResNet",):
        p = os.path.join(root, d)
        if p not in sys.path:
            sys.path.append(p)
    import resnet as rn
    from maai_hip import kernels as K
    torch.manual_seed(3)
    for (inp, planes, stride, hw) in ((256, 128, 2, 28), (512, 128, 1, 28)):
        ds = None
        if stride != 1 or inp != planes * 4:
            ds = torch.nn.Sequential(rn.conv1x1(inp, planes * 4, stride), torch.nn.BatchNorm2d(planes * 4))
        blk = rn.Bottleneck(inp, planes, stride, ds).cuda().train()
        x = torch.relu(torch.randn(8, hw, hw, inp)).cuda().bfloat16()
        outs = {}
        for fold in (False, True):
            E.set_fold(fold)
            out, recs, _, _ = E._block_fwd(blk, x, torch.bfloat16, True)
            gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).cuda().bfloat16()
            gm = E.relu_mask_grad(gout, out)
            grads = {}
            dx, _ = E.block_bwd(("block",) + recs, gm, grads, torch.bfloat16, prev=None, mask_input=False)
            torch.cuda.synchronize()
            outs[fold] = (dx, {n: grads[id(p)] for n, p in blk.named_parameters()})
        a, b = outs[True], outs[False]
        for nm, f, un in [("dx", a[0], b[0])] + [(n, a[1][n], b[1][n]) for n in a[1]]:
            f, un = f.double().reshape(-1), un.double().reshape(-1)
            cos = torch.nn.functional.cosine_similarity(f, un, dim=0).item()
            assert cos > 0.9995, (inp, planes, nm, cos)
            assert abs(f.norm().item() / un.norm().item() - 1) < 1e-2, (inp, planes, nm)
