"""Same-process A/B of the single- vs double-buffered 3x3 patch weight-gradient kernel (MAAI_WGRAD_PATCH_DB read per
call), plain and with the input normalised on load, over split-K budgets; results must agree to fp32 atomics order."""
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


os.environ["MAAI_WGRAD_PATCH"] = "1"
for (cin, cout, hw) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
    xs, xt = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    wq = torch.empty(cout, 3, 3, cin, device="cuda", dtype=torch.bfloat16)
    d = K.make_desc(x, wq, 1, 1, 1)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    for xf in (False, True):
        out, ref = [], {}
        for db in ("0", "1"):
            os.environ["MAAI_WGRAD_PATCH_DB"] = db
            for target in (256, 512, 768, 1536, 3072):
                dw = torch.zeros(cout, 3, 3, cin, device="cuda", dtype=torch.float32)
                if xf:
                    fn = lambda: check(lib().maai_conv2d_wgrad_xf(C.byref(d), K._p(x), K._p(dy), K._p(dw), BF16, target, K._p(xs), K._p(xt), 1, K._stream()), "wgrad")
                else:
                    fn = lambda: check(lib().maai_conv2d_wgrad_tuned(C.byref(d), K._p(x), K._p(dy), K._p(dw), BF16, target, K._stream()), "wgrad")
                dw.zero_()
                fn()
                torch.cuda.synchronize()
                ref.setdefault("r", dw.clone())
                err = float((dw - ref["r"]).abs().max() / ref["r"].abs().max())
                ms = timeit(fn)
                out.append("db%s/t%d %.3f ms %.0f TF/s%s" % (db, target, ms, fl / ms / 1e9, "" if err < 1e-4 else " ERR %.1e" % err))
        print("cin%4d hw%4d %s: " % (cin, hw, "xf   " if xf else "plain") + "  ".join(out), flush=True)
    del x, dy
