#!/usr/bin/env python3
"""Streaming pointwise kernel (conv_pws.hip) against the ring kernel (conv_igemm.h) on the benchmark's pointwise layer
shapes: bit-identical outputs, statistics to fp32 rounding, interleaved timings.   python scripts/pws_ab.py [B]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = 5


ARMS = {"ring": {"MAAI_CONV_PWS": "0"}, "s128": {"MAAI_CONV_PWS": "2", "MAAI_PWS_BN": "128"}, "s64": {"MAAI_CONV_PWS": "2", "MAAI_PWS_BN": "64"}}


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
    return {k: statistics.median(v) for k, v in res.items()}


def case(hw, cin, cout, lazy):
    n = B
    x = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, 1, 1, cin, device="cuda") / cin ** 0.5).bfloat16()
    s, t = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    inp = K.Lazy(x, s, t, True) if lazy else x
    fn = lambda: K.conv2d(inp, w, 1, 0, 0, stats=True)
    y0, st0 = run("ring", fn)
    same, serr = True, 0.0
    for arm in ("s128", "s64") * 3:
        y1, st1 = run(arm, fn)
        torch.cuda.synchronize()
        same &= torch.equal(y0, y1)
        # the slabs have one row per tile (the two kernels may tile differently): compare the column totals
        t0s, t1s = st0.double().sum(0), st1.double().sum(0)
        serr = max(serr, float(((t0s - t1s).abs() / (t0s.abs() + 1e-6 * t0s.abs().max())).max()))
    r = timeit(fn)
    by = 2.0 * n * hw * hw * (cin + cout)
    print("%4d->%4d @%3d %s  " % (cin, cout, hw, "lazy " if lazy else "plain") +
          "  ".join("%s %6.3f ms (%4.0f GB/s)" % (k, v, by / v / 1e6) for k, v in r.items()) +
          "   best stream %+5.1f %%   y identical: %s  stats rel %.1e" % ((r["ring"] / min(r["s128"], r["s64"]) - 1) * 100, same, serr), flush=True)
    return same


def join_case(hw, cin, cout, proj):
    n = B
    y3 = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    sc = torch.randn(n, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, 1, 1, cin, device="cuda") / cin ** 0.5).bfloat16()
    s, t = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    lazy = K.Lazy(y3, s, t, True, sc, s + 0.1, t - 0.1) if proj else K.Lazy(y3, s, t, True, sc.clamp_min(0))
    fn = lambda: K.conv2d(lazy, w, stats=True, join_out=True, join_bits=True)
    r0 = run("ring", fn)
    same = True
    for arm in ("s128", "s64") * 2:
        r1 = run(arm, fn)
        torch.cuda.synchronize()
        same &= torch.equal(r0[0], r1[0]) and torch.equal(r0[2], r1[2]) and torch.equal(r0[3], r1[3])
    r = timeit(fn)
    by = 2.0 * n * hw * hw * (3 * cin + cout)
    print("join %4d->%4d @%3d %s  " % (cin, cout, hw, "proj" if proj else "iden") +
          "  ".join("%s %6.3f ms (%4.0f GB/s)" % (k, v, by / v / 1e6) for k, v in r.items()) +
          "   best stream %+5.1f %%   identical: %s" % ((r["ring"] / min(r["s128"], r["s64"]) - 1) * 100, same), flush=True)
    return same


def chain_case(hw, cout, proj):
    """block boundary of layer 1: conv3 (64->256, lazy input) + join + next conv1, unchained vs chained"""
    n = B
    a2 = torch.randn(n, hw, hw, 64, device="cuda").bfloat16()
    w3 = (torch.randn(256, 1, 1, 64, device="cuda") / 8).bfloat16()
    w1 = (torch.randn(cout, 1, 1, 256, device="cuda") / 16).bfloat16()
    sc = torch.randn(n, hw, hw, 256, device="cuda").bfloat16()
    ps, pt = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    s3, t3 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.3
    src = K.Lazy(a2, ps, pt, True)
    extra = (sc, s3 + 0.1, t3 - 0.1) if proj else (sc.clamp_min(0),)
    os.environ.update(ARMS["ring"])
    os.environ["MAAI_CONV_PWS"] = "1"
    y3, _ = K.conv2d(src, w3, stats=True)

    def ev(fn, rounds=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)
    t_conv3 = ev(lambda: K.conv2d(src, w3, stats=True))
    t_join = ev(lambda: K.conv2d(K.Lazy(y3, s3, t3, True, *extra), w1, stats=True, join_out=True, join_bits=True))
    t_stats = ev(lambda: K.conv2d_stats_only(src, w3))
    lz = K.Lazy(None, s3, t3, True, *extra, pre=(src, w3))
    t_ch0 = ev(lambda: K.conv2d_chained(lz, w1, stats=True, join_bits=False, keep_y=False))
    t_ch1 = ev(lambda: K.conv2d_chained(lz, w1, stats=True, join_bits=True, keep_y=True))
    print("chain 64->256->%3d @%3d %s: conv3 %.3f + join %.3f = %.3f ms | stats-only %.3f + chain %.3f = %.3f (no backward) ; + chain(keep y, bits) %.3f = %.3f" %
          (cout, hw, "proj" if proj else "iden", t_conv3, t_join, t_conv3 + t_join, t_stats, t_ch0, t_stats + t_ch0, t_ch1, t_stats + t_ch1), flush=True)


def bwd3_case(hw):
    """conv3 + bn3 backward of a 64 -> 256 bottleneck: apply-on-load data gradient + weight gradient vs the fused launch"""
    n = B
    g = torch.randn(n, hw, hw, 256, device="cuda").bfloat16()
    y3 = torch.randn(n, hw, hw, 256, device="cuda").bfloat16()
    y2 = torch.randn(n, hw, hw, 64, device="cuda").bfloat16()
    wd = (torch.randn(64, 1, 1, 256, device="cuda") / 16).bfloat16()
    k1, k2, k3 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.1, torch.randn(256, device="cuda") * 0.1
    mean2, s2, t2 = torch.zeros(64, device="cuda"), torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    m = n * hw * hw
    dx = torch.empty_like(y2)
    part = torch.zeros((m + 127) // 128, 2, 64, device="cuda")
    dz = torch.empty_like(g)
    lz = K.Lazy(y2, s2, t2, True)

    def ev(fn, rounds=5):
        fn()
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)
    t_d = ev(lambda: K.conv2d_store_reduce(g, wd, 1, 0, 0, dx, part, y2, mean2, s2, t2, None, axf=(y3, k1, k2, k3, dz)))
    t_w = ev(lambda: K.conv2d_wgrad(lz, dz, 1, 1, 1, 0, 0))
    t_f = ev(lambda: K.conv_bwd3(g, y3, y2, wd, k1, k2, k3, mean2, s2, t2))
    by = 2.0 * m * (256 * 2 + 64 * 2)
    print("bwd3 @%3d: apply-on-load data gradient %.3f + weight gradient %.3f = %.3f ms | fused %.3f ms (%4.0f GB/s)" %
          (hw, t_d, t_w, t_d + t_w, t_f, by / t_f / 1e6), flush=True)


if __name__ == "__main__":
    print("B = %d" % B)
    ok = True
    if len(sys.argv) > 2 and sys.argv[2] == "bwd3":
        bwd3_case(224)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "chain":
        for hw, cout, proj in ((224, 64, False), (224, 128, False), (224, 64, True)):
            chain_case(hw, cout, proj)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "join":
        for hw, cin, cout, proj in ((224, 256, 64, False), (224, 256, 128, False), (224, 256, 64, True), (112, 256, 128, False)):
            ok &= join_case(hw, cin, cout, proj)
        print("all identical" if ok else "MISMATCH")
        sys.exit(0 if ok else 1)
    for lazy in (False, True):
        for hw, cin, cout in ((224, 64, 64), (224, 64, 256), (112, 128, 512), (112, 128, 128), (56, 256, 1024), (56, 256, 256), (56, 256, 512),
                              (224, 256, 64), (112, 256, 128)):
            ok &= case(hw, cin, cout, lazy)
    print("all identical" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
