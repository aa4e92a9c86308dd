"""What bounds the expanding pointwise layers?  Time vs statistics on/off, output width, K."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K


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


B, hw = 256, 56
for cin, cout in ((256, 1024), (256, 512), (256, 256), (256, 2048), (512, 1024), (128, 1024), (64, 1024)):
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, 1, 1, cin, device="cuda") / cin ** 0.5).to(torch.bfloat16)
    m = B * hw * hw
    out = torch.empty((B, hw, hw, cout), dtype=torch.bfloat16, device="cuda")
    t1 = timeit(lambda: K.conv2d(x, w, 1, 0, 0, stats=True, out=out))
    t0 = timeit(lambda: K.conv2d(x, w, 1, 0, 0, stats=False, out=out))
    fl = 2.0 * m * cin * cout
    by = 2.0 * m * (cin + cout)
    print("cin%5d cout%5d: stats %.3f ms  no-stats %.3f ms   %.0f TF/s  %.0f GB/s   per 128x128 tile %.2f us" % (
        cin, cout, t1, t0, fl / t1 / 1e9, by / t1 / 1e6, t1 * 1e3 / ((m / 128) * (cout / 128)) * 256 * 3), flush=True)
