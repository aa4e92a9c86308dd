"""Timing of the Gram passes at the step's shapes (256 images), HIP events: python3 scripts/gram_ab.py  (MAAI_LIB_PATH=<other build> for A/B)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-active-ai_amd"))
import torch
from maai_hip import kernels as K

for (m, c, xf) in ((12845056, 64, True), (3211264, 128, False), (802816, 256, False), (200704, 512, False)):
    y = torch.randn(1, 1, m, c, device="cuda").bfloat16()
    x = K.Lazy(y, torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.5, True) if xf else y
    for det in (True, False):
        ms = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            (K.gram_deterministic if det else K.gram)(x)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        print("M %d C %d xf %d %s: %.3f ms  (%.2f TB/s, %.0f TFLOP/s)" % (m, c, xf, "deterministic" if det else "atomic", min(ms[2:]), 2e-9 * m * c / min(ms[2:]), 2e-9 * m * c * c / min(ms[2:])), flush=True)
