"""Same-process A/B of 128- vs 256-row tiles on the pointwise layers (MAAI_CONV_BM read per call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
B = int(os.environ.get("B", "256"))
shapes = [(256, 1024, 56), (1024, 256, 56), (128, 512, 112), (512, 128, 112), (512, 2048, 28), (2048, 512, 28), (64, 256, 224), (256, 128, 224)]


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


for (cin, cout, hw) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, 1, 1, cin, device="cuda") / cin ** 0.5).to(torch.bfloat16)
    fn = lambda: K.conv2d(x, w, 1, 0, 0, stats=True)
    m = B * hw * hw
    fl = 2.0 * m * cout * cin
    by = 2.0 * m * (cin + cout)
    res = {}
    for rep in range(2):
        for bm in ("128", "256"):
            os.environ["MAAI_CONV_BM"] = bm
            res.setdefault(bm, []).append(timeit(fn))
    print("cin%5d cout%5d hw%4d : " % (cin, cout, hw) + "  ".join(
        "bm%s %.3f ms %.0f TF/s %.0f GB/s" % (bm, min(v), fl / min(v) / 1e9, by / min(v) / 1e6) for bm, v in res.items()), flush=True)
    del x
