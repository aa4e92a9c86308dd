"""Micro-benchmark of maai_conv2d_igemm / wgrad on the ResNet-50@224 layer shapes (B=64)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
B = int(os.environ.get("B", "64"))
shapes = [  # cin, cout, hw, k, stride
    (64, 256, 224, 1, 1), (256, 64, 224, 1, 1), (64, 64, 224, 3, 1), (128, 512, 112, 1, 1), (512, 128, 112, 1, 1), (128, 128, 112, 3, 1),
    (256, 1024, 56, 1, 1), (1024, 256, 56, 1, 1), (256, 256, 56, 3, 1), (512, 2048, 28, 1, 1), (2048, 512, 28, 1, 1), (512, 512, 28, 3, 1)]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
res = {}
for (cin, cout, hw, k, s) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16)
    p = k // 2
    if which == "fwd":
        fn = lambda: K.conv2d(x, w, s, p, p, stats=os.environ.get("STATS", "1") == "1")
    else:
        dy = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
        fn = lambda: K.conv2d_wgrad(x, dy, k, k, s, p, p)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    m = B * hw * hw
    fl = 2.0 * m * cout * k * k * cin
    by = 2.0 * (m * cin + m * cout)
    print("%s cin%5d cout%5d hw%4d k%d : %7.3f ms  %7.1f TF/s  %7.1f GB/s" % (which, cin, cout, hw, k, ms, fl / ms / 1e9, by / ms / 1e6), flush=True)
