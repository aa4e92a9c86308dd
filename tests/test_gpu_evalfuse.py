"""Inference with frozen BatchNorm statistics (SimCLR.py: the backbone in eval mode; SURVEY 8-f1 linear probe, f4 frozen
backbone): conv + BN (+ shortcut) + ReLU in ONE launch (MAAI_EPI_BN_ACT) on the streaming, ping-pong and halo kernels —
round 3; the ring kernel's 128-row tile had it before.  The epilogue applies maai_bn_act_fwd's arithmetic to the
bf16-rounded tile, so every fused launch must be BIT-IDENTICAL to the plain launch followed by the BatchNorm pass; and the
engine's eval forward with the rule on must equal the one with it off.  Reference semantics: resnet.py:101-110, 118-133
with nn.BatchNorm2d in eval mode."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# name, env that puts the fused launch on the kernel under test, (N, H, W, Cin, Cout, k, stride), lazy input
CASES = [
    ("streaming 64->256", dict(MAAI_CONV_PWS="2"), (3, 21, 19, 64, 256, 1, 1), False),
    ("streaming 128->512 lazy", dict(MAAI_CONV_PWS="2"), (2, 28, 28, 128, 512, 1, 1), True),
    ("streaming 256->1024", dict(MAAI_CONV_PWS="2"), (2, 14, 15, 256, 1024, 1, 1), False),
    ("streaming 256->64 lazy", dict(MAAI_CONV_PWS="2"), (2, 17, 16, 256, 64, 1, 1), True),
    ("ping-pong 3x3 256", dict(MAAI_CONV_PP="2"), (3, 14, 14, 256, 256, 3, 1), False),
    ("ping-pong 3x3 256 stride 2", dict(MAAI_CONV_PP="2"), (2, 28, 28, 256, 256, 3, 2), False),
    ("ping-pong 1x1 1024->256", dict(MAAI_CONV_PP="2"), (2, 15, 14, 1024, 256, 1, 1), False),
    ("ping-pong 1x1 512->2048", dict(MAAI_CONV_PP="2"), (3, 7, 7, 512, 2048, 1, 1), False),
    ("ping-pong persistent 1x1 1024->256", dict(MAAI_CONV_PP="2"), (8, 96, 96, 1024, 256, 1, 1), False),
    ("ping-pong 3x3 512 ragged", dict(MAAI_CONV_PP="2"), (1, 9, 7, 512, 512, 3, 1), False),
    ("halo 3x3 128", dict(MAAI_CONV_HALO="1", MAAI_CONV_PP="0"), (2, 30, 32, 128, 128, 3, 1), False),
    ("halo 3x3 64", dict(MAAI_CONV_HALO="1", MAAI_CONV_C64="0"), (2, 33, 20, 64, 64, 3, 1), False),
    ("ring 128-row 512->2048", dict(MAAI_CONV_PP="0"), (2, 7, 7, 512, 2048, 1, 1), False),
    ("ring 128 x 256 tile 512->2048", dict(MAAI_CONV_PP="0", MAAI_CONV_BN="256"), (3, 9, 9, 512, 2048, 1, 1), False),
    ("ring 256-row 512->128", dict(MAAI_CONV_PP="0", MAAI_CONV_BM="256"), (3, 20, 21, 512, 128, 1, 1), False),
]


@pytest.mark.parametrize("res", [False, True], ids=["plain", "residual"])
@pytest.mark.parametrize("relu", [True, False], ids=["relu", "linear"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_fused_epilogue_equals_launch_plus_pass(K, case, relu, res):
    name, ev, (n, h, w_, cin, cout, k, stride), lazy = case
    g = torch.Generator().manual_seed(len(name) * 131 + cin)
    x = torch.randn(n, h, w_, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).cuda().bfloat16()
    scale = ((torch.rand(cout, generator=g) + 0.5) * torch.where(torch.rand(cout, generator=g) < 0.2, -1.0, 1.0)).cuda()
    shift = (torch.randn(cout, generator=g) * 0.5).cuda()
    pad = k // 2
    oh, ow = (h + 2 * pad - k) // stride + 1, (w_ + 2 * pad - k) // stride + 1
    r = torch.randn(n, oh, ow, cout, generator=g).cuda().bfloat16() if res else None
    inp = x
    if lazy:
        xs, xt = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.3).cuda()
        inp = K.Lazy(x, xs, xt, True)
    with env(**ev):
        y = K.conv2d(inp, w, stride, pad, pad)
        ref = K.bn_act_fwd(y, scale, shift, r, relu)
        got = K.conv2d_bn_act(inp, w, scale, shift, r, relu, stride, pad, pad)
        got2 = K.conv2d_bn_act(inp, w, scale, shift, r, relu, stride, pad, pad)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert torch.equal(got, ref), "%s: fused epilogue differs from launch + pass (max %.4g)" % (name, float((got.float() - ref.float()).abs().max()))
    assert torch.equal(got, got2)
    assert float(ref.float().abs().max()) > 0.1


def test_fast_rule(K):
    """which eval units fuse beyond the small-tensor limit: the benchmark's shapes"""
    import torch.nn as nn
    bf = torch.bfloat16
    c = nn.Conv2d(256, 1024, 1, bias=False)
    assert K.conv_bn_act_fast(c, 256, 56, 56, bf) and K.conv_bn_act_fast(c, 256, 56, 56, bf, lazy=True)   # streaming
    c = nn.Conv2d(256, 256, 3, padding=1, bias=False)
    assert K.conv_bn_act_fast(c, 256, 56, 56, bf) and not K.conv_bn_act_fast(c, 256, 56, 56, bf, lazy=True)   # ping-pong: tensors only
    c = nn.Conv2d(128, 128, 3, padding=1, bias=False)
    assert K.conv_bn_act_fast(c, 256, 112, 112, bf)    # halo
    c = nn.Conv2d(512, 128, 1, bias=False)              # channel-reducing ring layer on 256-row tiles
    assert K.conv_bn_act_fast(c, 256, 112, 112, bf)


@pytest.mark.parametrize("arch,size,batch", [("resnet50", 64, 8), ("resnet18", 64, 8), ("resnet50", 224, 4)])
def test_eval_forward_with_fast_rule_equals_pass_path(arch, size, batch):
    """eval forward (frozen statistics): rule on == rule off, bit for bit, and fewer BatchNorm passes — bottleneck and basic
    blocks, and at 224^2 the shapes whose launches take the streaming / chained / persistent kernels"""
    PKG = os.path.join(ROOT, "multimodal-active-ai_amd")
    SIM = os.path.join(PKG, "SimCLR")
    for d in (PKG, SIM, os.path.join(SIM, "ResNet"), os.path.join(SIM, "MLP")):
        if d not in sys.path:
            sys.path.insert(0, d)
    import resnet as rn
    from maai_hip import engine, kernels as K
    torch.manual_seed(3)
    f = getattr(rn, arch)(crop_measures=1).cuda()
    for m in f.modules():   # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    f.eval()
    x = torch.randint(0, 256, (batch, 3, size, size), device="cuda").float()
    dtype = torch.bfloat16
    outs, passes = [], []
    old = dict(engine._EVAL_FUSE)
    try:
        for fast in (False, True):
            engine._EVAL_FUSE["fast"] = fast
            engine._EVAL_FUSE["max_rows"] = 0      # (at this size everything would fuse by the small-tensor rule)
            with torch.no_grad(), K.profile() as prof:
                o, _ = engine.backbone_fwd(f, x, dtype, keep=False)
            outs.append(o.clone())
            passes.append(sum(v["launches"] for k, v in prof.table().items() if k.startswith("bn_act_fwd")))
    finally:
        engine._EVAL_FUSE.update(old)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    assert passes[1] < passes[0], passes


@pytest.mark.parametrize("size,batch", [(64, 8), (128, 4)])
def test_nograd_training_forward_with_fused_units_equals_unfused(size, batch):
    """The no-grad view of the SimCLR step (training-mode BatchNorm, nothing kept for a backward): conv3 of stages 2-3 as a
    statistics-only launch + a BatchNorm-epilogue launch (engine._FUSE nograd rule) == launch + join, bit for bit — features
    and running statistics"""
    PKG = os.path.join(ROOT, "multimodal-active-ai_amd")
    SIM = os.path.join(PKG, "SimCLR")
    for d in (PKG, SIM, os.path.join(SIM, "ResNet"), os.path.join(SIM, "MLP")):
        if d not in sys.path:
            sys.path.insert(0, d)
    import copy
    import resnet as rn
    from maai_hip import engine, kernels as K
    torch.manual_seed(5)
    f0 = rn.resnet50(crop_measures=1).cuda()
    for m in f0.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    x = torch.randint(0, 256, (batch, 3, size, size), device="cuda").float()
    outs, stats, launches = [], [], []
    old = dict(engine._FUSE)
    engine.set_gram_stats(False)   # (bit-identity with launch + join needs the statistics-only launch's slab, not the Gram form)
    try:
        for lim in (0, 256):
            engine._FUSE["nograd_max_cin"] = lim
            f = copy.deepcopy(f0)
            f.train()
            with torch.no_grad(), K.profile() as prof:
                o, _ = engine.backbone_fwd(f, x, torch.bfloat16, keep=False)
            outs.append(o.clone())
            stats.append([b.clone() for n, b in f.named_buffers()])
            launches.append(sum(v["launches"] for k, v in prof.table().items() if k.startswith("bn_act_fwd")))
    finally:
        engine._FUSE.update(old)
        engine.set_gram_stats(True)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    for a, b in zip(stats[0], stats[1]):
        assert torch.equal(a, b)
    assert launches[1] < launches[0], launches
