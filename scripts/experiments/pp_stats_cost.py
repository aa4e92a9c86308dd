"""What does the statistics epilogue of the ping-pong kernel cost?  conv2d(stats=True) vs conv2d(stats=False), 256 images."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
K.AUTOTUNE[0] = False
def t(fn, n=12):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
for (hw, cin, cout, k) in ((56, 256, 256, 3), (28, 512, 512, 3), (56, 1024, 256, 1), (28, 2048, 512, 1), (112, 512, 256, 1)):
    x = torch.randn(256, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, k, k, cin, device="cuda") / (k * k * cin) ** 0.5).bfloat16()
    y = torch.empty(256, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
    a = t(lambda: K.conv2d(x, w, 1, k // 2, k // 2, stats=True, out=y))
    b = t(lambda: K.conv2d(x, w, 1, k // 2, k // 2, stats=False, out=y))
    a2 = t(lambda: K.conv2d(x, w, 1, k // 2, k // 2, stats=True, out=y))
    b2 = t(lambda: K.conv2d(x, w, 1, k // 2, k // 2, stats=False, out=y))
    print("%4d->%4d k%d @%3d  stats %.3f / %.3f ms   no stats %.3f / %.3f ms" % (cin, cout, k, hw, a, a2, b, b2), flush=True)
    del x, w, y
