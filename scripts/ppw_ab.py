#!/usr/bin/env python3
"""Ping-pong weight-gradient kernel (conv_ppw.hip) against the ring / patch kernels on the benchmark's MFMA-bound layer
shapes, interleaved timings.   python scripts/ppw_ab.py [B]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = 7
ARMS = {"old": {"MAAI_WGRAD_PP": "0"}, "pp": {"MAAI_WGRAD_PP": "2"}}
K.AUTOTUNE[0] = False


def run(arm, fn):
    os.environ.update(ARMS[arm])
    return fn()


def timeit(fn):
    res = {k: [] for k in ARMS}
    for p in ARMS:
        run(p, fn)
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for p in ARMS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(p, fn)
            e1.record()
            e1.synchronize()
            res[p].append(e0.elapsed_time(e1))
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def case(hw, cin, cout, k, stride=1):
    n = B
    pad = k // 2
    oh = (hw + 2 * pad - k) // stride + 1
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    dy = (torch.randn(n, oh, oh, cout, device="cuda") * 0.05).bfloat16()
    fn = lambda: K.conv2d_wgrad(x, dy, k, k, stride, pad, pad)
    d0 = run("old", fn)
    d1 = run("pp", fn)
    torch.cuda.synchronize()
    err = float((d0 - d1).abs().max() / d0.abs().max())
    r = timeit(fn)
    fl = 2.0 * n * oh * oh * cout * k * k * cin
    print("wgrad %4d->%4d k%d s%d @%3d  old %.3f ms (%6.0f TF)  pp %.3f ms (%6.0f TF)  min %.3f / %.3f  rel diff %.1e" % (
        cin, cout, k, stride, hw, r["old"][0], fl / r["old"][0] / 1e9, r["pp"][0], fl / r["pp"][0] / 1e9, r["old"][1], r["pp"][1], err), flush=True)
    del x, dy


if __name__ == "__main__":
    case(56, 256, 256, 3)
    case(28, 512, 512, 3)
    case(112, 256, 256, 3, stride=2)
    case(56, 512, 512, 3, stride=2)
    case(56, 256, 1024, 1)
    case(56, 1024, 256, 1)
    case(28, 512, 2048, 1)
    case(28, 2048, 512, 1)
    case(56, 1024, 512, 1)
    case(56, 512, 1024, 1, stride=2)
    case(28, 1024, 2048, 1, stride=2)
    case(112, 256, 512, 1, stride=2)
    case(56, 256, 512, 1)
