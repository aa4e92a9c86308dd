"""One kernel shape, a few launches (for rocprofv3 --pmc passes):
   python scripts/conv_one.py CIN COUT HW K B [fwd|wgrad|bwd3|chain]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
cin, cout, hw, k, B = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else "fwd"
K.AUTOTUNE[0] = False
if mode == "fwd":
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16)
    fn = lambda: K.conv2d(x, w, 1, k // 2, k // 2, stats=True)
elif mode == "wgrad":
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    dy = (torch.randn(B, hw, hw, cout, device="cuda") * 0.05).to(torch.bfloat16)
    fn = lambda: K.conv2d_wgrad(x, dy, k, k, 1, k // 2, k // 2)
elif mode == "bwd3":   # the fused backward of a 64 -> 256 bottleneck's last unit at this plane size
    g = torch.randn(B, hw, hw, 256, device="cuda").to(torch.bfloat16)
    y3 = torch.randn(B, hw, hw, 256, device="cuda").to(torch.bfloat16)
    y2 = torch.randn(B, hw, hw, 64, device="cuda").to(torch.bfloat16)
    wd = (torch.randn(64, 1, 1, 256, device="cuda") / 16).to(torch.bfloat16)
    k1, k2, k3 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1, torch.randn(256, device="cuda") * 0.1
    m2, s2, t2 = torch.randn(64, device="cuda") * 0.1, torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    fn = lambda: K.conv_bwd3(g, y3, y2, wd, k1, k2, k3, m2, s2, t2)
elif mode == "chain":  # the chained block boundary of layer 1 (no backward pass follows)
    a2 = torch.randn(B, hw, hw, 64, device="cuda").to(torch.bfloat16)
    w3 = (torch.randn(256, 1, 1, 64, device="cuda") / 8).to(torch.bfloat16)
    w1 = (torch.randn(64, 1, 1, 256, device="cuda") / 16).to(torch.bfloat16)
    sc = torch.randn(B, hw, hw, 256, device="cuda").clamp_min(0).to(torch.bfloat16)
    s3, t3 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.5
    fn = lambda: K.conv2d_chained(K.Lazy(None, s3, t3, True, sc, pre=(a2, w3)), w1, stats=True, join_bits=False)
else:
    raise SystemExit(__doc__)
for _ in range(3):
    fn()
torch.cuda.synchronize()
