"""Same-process A/B of conv variants selected by MAAI_CONV_FLAGS (read per call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
B = 64
shapes = [(64, 256, 224, 1), (256, 64, 224, 1), (128, 512, 112, 1), (256, 1024, 56, 1), (512, 2048, 28, 1), (256, 256, 56, 3)]
flags = sys.argv[1:] or ["0", "1"]
for (cin, cout, hw, k) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16)
    p = k // 2
    res = {f: [] for f in flags}
    for rnd in range(4):
        for f in flags:
            os.environ["MAAI_CONV_FLAGS"] = f
            for _ in range(2):
                K.conv2d(x, w, 1, p, p, stats=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                K.conv2d(x, w, 1, p, p, stats=True)
            e1.record(); torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / 5)
    print("cin%5d cout%5d hw%4d k%d : " % (cin, cout, hw, k) + "  ".join("flags=%s min %.3f med %.3f ms" % (f, min(v), sorted(v)[len(v) // 2]) for f, v in res.items()), flush=True)
