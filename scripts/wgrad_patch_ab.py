"""Same-process A/B of the ring vs patch-staged 3x3 weight-gradient kernels (MAAI_WGRAD_PATCH read per call)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
from maai_hip._lib import lib, check, BF16
B = int(os.environ.get("B", "256"))
shapes = [(64, 64, 224), (128, 128, 112), (256, 256, 56), (512, 512, 28)]


def timeit(fn, n=5):
    for _ in range(2):
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
    dy = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
    wq = torch.empty(cout, 3, 3, cin, device="cuda", dtype=torch.bfloat16)
    d = K.make_desc(x, wq, 1, 1, 1)
    dw = torch.zeros(cout, 3, 3, cin, device="cuda", dtype=torch.float32)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    out = []
    for mode in ("0", "1"):
        os.environ["MAAI_WGRAD_PATCH"] = mode
        for target in ((768, 1536, 3072) if mode == "0" else (768, 1536, 3072, 6144)):
            fn = lambda: check(lib().maai_conv2d_wgrad_tuned(C.byref(d), K._p(x), K._p(dy), K._p(dw), BF16, target, K._stream()), "wgrad")
            ms = timeit(fn)
            out.append("p%s/t%d %.3f ms %.0f TF/s" % (mode, target, ms, fl / ms / 1e9))
    print("cin%4d cout%4d hw%4d : " % (cin, cout, hw) + "  ".join(out), flush=True)
    del x, dy
