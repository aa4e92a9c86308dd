"""Same-process A/B of row-staged vs halo-staged 3x3 kernels (MAAI_CONV_HALO is read per call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
B = int(os.environ.get("B", "64"))
shapes = [(64, 64, 224), (128, 128, 112), (256, 256, 56), (512, 512, 28), (64, 64, 32), (128, 128, 16)]


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
    b = B if hw > 32 else B * 16
    x = torch.randn(b, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, 3, 3, cin, device="cuda") / (cin * 9) ** 0.5).to(torch.bfloat16)
    fn = lambda: K.conv2d(x, w, 1, 1, 1, stats=True)
    fl = 2.0 * b * hw * hw * cout * 9 * cin
    res = {}
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["MAAI_CONV_HALO"] = mode
            res.setdefault(mode, []).append(timeit(fn))
    print("B%4d cin%4d cout%4d hw%4d : " % (b, cin, cout, hw) + "  ".join(
        "halo=%s %.3f ms %.0f TF/s" % (m, min(v), fl / min(v) / 1e9) for m, v in res.items()), flush=True)
