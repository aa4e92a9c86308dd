#!/usr/bin/env python3
"""Time csrc/conv_dfold.hip at the benchmark's layer-1 shape (256 images x 224^2): MAAI_DFOLD_RESIDENT=0|1 python scripts/dfold_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K  # noqa: E402

m = 256 * 224 * 224
g = torch.Generator(device="cuda").manual_seed(1)
gg = torch.randn(1, 1, m, 256, device="cuda", generator=g).bfloat16()
y2 = torch.randn(1, 1, m, 64, device="cuda", generator=g).bfloat16()
wcat = (torch.randn(64, 1, 1, 320, device="cuda", generator=g) / 16).bfloat16()
cn, mean2, t2 = (torch.randn(64, device="cuda", generator=g) * 0.1 for _ in range(3))
s2 = torch.rand(64, device="cuda", generator=g) + 0.5
dg = torch.randn(64, device="cuda", generator=g) * 0.3
for acc in (False, True):
    dx = torch.zeros(1, 1, m, 64, device="cuda", dtype=torch.bfloat16) if acc else None
    for _ in range(3):
        K.conv_dfold(gg, y2, wcat, cn, mean2, s2, t2, dx=dx, dg=dg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        K.conv_dfold(gg, y2, wcat, cn, mean2, s2, t2, dx=dx, dg=dg)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    gb = 2 * (m * 256 + m * 64 * (3 if acc else 2)) / 1e9
    print("dfold resident=%s acc=%d: %.3f ms  %.2f TB/s" % (os.environ.get("MAAI_DFOLD_RESIDENT", "1"), int(acc), ms, gb / ms))
