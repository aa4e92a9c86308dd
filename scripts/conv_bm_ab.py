"""Same-process A/B of the implicit-GEMM tile rows (MAAI_CONV_BM is read per call) on the spatial ResNet-50 layers."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
B = int(os.environ.get("B", "64"))
shapes = [(64, 64, 224, 3, 1), (128, 128, 112, 3, 1), (256, 256, 56, 3, 1), (512, 512, 28, 3, 1), (128, 128, 224, 3, 2), (256, 256, 112, 3, 2)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (cin, cout, hw, k, s) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16)
    p = k // 2
    fn = lambda: K.conv2d(x, w, s, p, p, stats=True)
    oh = (hw + 2 * p - k) // s + 1
    fl = 2.0 * B * oh * oh * cout * k * k * cin
    res = {}
    for rep in range(2):
        for bm in ("128", "256"):
            os.environ["MAAI_CONV_BM"] = bm
            res.setdefault(bm, []).append(timeit(fn))
    print("cin%4d cout%4d hw%4d k%d s%d : " % (cin, cout, hw, k, s) + "  ".join(
        "bm%s %.3f ms %.0f TF/s" % (bm, min(v), fl / min(v) / 1e9) for bm, v in res.items()), flush=True)
