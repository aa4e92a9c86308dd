import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-active-ai_amd")
for d in (ROOT, PKG):
    sys.path.insert(0, d)
from maai_hip import kernels as K, engine

def nhwc(x, dt): return x.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
def back(y): return y.float().cpu().permute(0, 3, 1, 2)
dt = torch.float32
g = torch.Generator().manual_seed(0)
B = 16
cases = [(64, 64, 32, 3, 1, 1), (64, 128, 32, 3, 2, 1), (64, 128, 32, 1, 2, 0), (128, 128, 16, 3, 1, 1), (128, 256, 16, 3, 2, 1),
         (128, 256, 16, 1, 2, 0), (256, 256, 8, 3, 1, 1), (256, 512, 8, 3, 2, 1), (256, 512, 8, 1, 2, 0), (512, 512, 4, 3, 1, 1)]
for (cin, cout, hw, k, s, p) in cases:
    x = torch.randn(B, cin, hw, hw, generator=g); w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, s, p); dy = torch.randn(y.shape, generator=g, dtype=torch.float64); y.backward(dy)
    yh = K.conv2d(nhwc(x, dt), nhwc(w, dt), s, p, p)
    e_f = ((back(yh).double() - y.detach()).abs().max() / y.detach().abs().max()).item()
    dyh = nhwc(dy.float(), dt)
    dw = K.conv2d_wgrad(nhwc(x, dt), dyh, k, k, s, p, p).cpu().permute(0, 3, 1, 2).double()
    e_w = ((dw - wd.grad).abs().max() / wd.grad.abs().max()).item()
    wp = torch.nn.Parameter(w.cuda())
    dx, _ = engine.conv_dgrad(dyh, wp, k, s, p, (hw, hw), dt)
    e_d = ((back(dx).double() - xd.grad).abs().max() / xd.grad.abs().max()).item()
    base = torch.randn(B, cin, hw, hw, generator=g)
    acc = nhwc(base, dt)
    engine.conv_dgrad(dyh, wp, k, s, p, (hw, hw), dt, out=acc, accumulate=True)
    e_a = ((back(acc).double() - (xd.grad + base.double())).abs().max() / xd.grad.abs().max()).item()
    print("cin %4d cout %4d hw %3d k %d s %d | fwd %.1e wgrad %.1e dgrad %.1e dgrad+acc %.1e" % (cin, cout, hw, k, s, e_f, e_w, e_d, e_a))
