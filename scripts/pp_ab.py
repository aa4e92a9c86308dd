#!/usr/bin/env python3
"""Ping-pong kernel (conv_pp.hip) against the ring / halo kernels (conv_igemm.h) on the benchmark's MFMA-bound layer
shapes: bit-identical outputs, interleaved timings.   python scripts/pp_ab.py [B]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = 7
ARMS = {"ring": {"MAAI_CONV_PP": "0", "MAAI_CONV_C64": "0"}, "pp": {"MAAI_CONV_PP": "2", "MAAI_CONV_C64": "1"}}


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


def case(hw, cin, cout, k, stride=1, dgrad=False, lazy=False):
    n = B
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, k, k, cin, device="cuda") / (k * k * cin) ** 0.5).bfloat16()
    pad = k // 2
    oh = (hw + 2 * pad - k) // stride + 1
    if dgrad:
        yb = torch.randn(n, oh, oh, cout, device="cuda").bfloat16()
        mean = torch.randn(cout, device="cuda") * 0.1
        s, t = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.3
        out = torch.empty_like(yb)

        def fn():
            rows = K.conv2d_stats_rows(x, w, 1, pad, pad)
            slab = torch.empty(rows, 2, cout, device="cuda")
            K.conv2d_store_reduce(x, w, 1, pad, pad, out, slab, yb, mean, s, t, None)
            return out.clone(), slab
    elif lazy:
        xs, xt = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
        fn = lambda: K.conv2d(K.Lazy(x, xs, xt, True), w, stride, pad, pad, stats=True)
    else:
        fn = lambda: K.conv2d(x, w, stride, pad, pad, stats=True)
    y0, st0 = run("ring", fn)
    y1, st1 = run("pp", fn)
    torch.cuda.synchronize()
    same = torch.equal(y0, y1)
    r = timeit(fn)
    fl = 2.0 * n * oh * oh * cout * k * k * cin
    print("%s %4d->%4d k%d s%d @%3d  ring %.3f ms (%6.0f TF)  pp %.3f ms (%6.0f TF)  min %.3f / %.3f  %s" % (
        "dgrad" if dgrad else "fwd  ", cin, cout, k, stride, hw, r["ring"][0], fl / r["ring"][0] / 1e9, r["pp"][0], fl / r["pp"][0] / 1e9,
        r["ring"][1], r["pp"][1], "identical" if same else "DIFFERENT"), flush=True)
    del x, w


if __name__ == "__main__":
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    if which in ("all", "c256"):
        case(56, 256, 256, 3)
        case(28, 512, 512, 3)
        case(56, 256, 256, 3, dgrad=True)
        case(28, 512, 512, 3, dgrad=True)
        case(56, 1024, 256, 1)
        case(56, 256, 1024, 1)
        case(28, 2048, 512, 1)
        case(28, 512, 2048, 1)
        case(56, 1024, 512, 1)
        case(112, 512, 256, 1)
        case(112, 256, 256, 3, stride=2)
        case(56, 512, 512, 3, stride=2)
        case(28, 1024, 2048, 1)
    if which in ("all", "c64"):
        case(224, 64, 64, 3)
        case(224, 64, 64, 3, lazy=True)
        case(224, 64, 64, 3, dgrad=True)
    if which in ("all", "c128", "new"):
        case(112, 128, 128, 3)
        case(112, 128, 128, 3, dgrad=True)
        case(224, 128, 128, 3, stride=2)
        case(112, 512, 128, 1)
        case(56, 1024, 256, 1, dgrad=True)
        case(28, 2048, 512, 1, dgrad=True)
        case(56, 512, 256, 1, dgrad=True)
        case(112, 512, 128, 1, dgrad=True)
