#!/usr/bin/env python3
"""In-process A/B of normalise-on-load against the materialised path at the benchmark's layer shapes (B images):
    materialised = BatchNorm pass + plain launch      vs      lazy = one launch on the raw tensor
for the forward convolutions, the residual join (next block's conv1) and the weight gradients.
Interleaved rounds, medians (cdna_hip_programming.md rule 24).   python scripts/xf_ab.py [B]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = 5


def timeit(fns):
    """fns: dict name -> callable; returns dict name -> median ms over interleaved rounds"""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            e1.synchronize()
            res[k].append(e0.elapsed_time(e1))
    return {k: statistics.median(v) for k, v in res.items()}


def coeffs(c):
    return (torch.rand(c, device="cuda") + 0.5), torch.randn(c, device="cuda") * 0.3


def fwd_case(name, hw, cin, cout, k, stride=1):
    pad = k // 2
    y = torch.randn(B, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).bfloat16()
    s, t = coeffs(cin)
    act = K.bn_act_fwd(y, s, t, None, True)
    r = timeit({"pass": lambda: K.bn_act_fwd(y, s, t, None, True),
                "plain": lambda: K.conv2d(act, w, stride, pad, pad, stats=True),
                "lazy": lambda: K.conv2d(K.Lazy(y, s, t, True), w, stride, pad, pad, stats=True)})
    print("fwd   %-28s pass %6.3f + plain %6.3f = %6.3f   lazy %6.3f   gain %+6.3f ms" %
          (name, r["pass"], r["plain"], r["pass"] + r["plain"], r["lazy"], r["pass"] + r["plain"] - r["lazy"]), flush=True)


def join_case(name, hw, cin, cout):
    y = torch.randn(B, hw, hw, cin, device="cuda").bfloat16()
    b = torch.randn(B, hw, hw, cin, device="cuda").bfloat16().clamp_min(0)
    w = (torch.randn(cout, 1, 1, cin, device="cuda") / cin ** 0.5).bfloat16()
    s, t = coeffs(cin)
    act, _ = K.bn_act_fwd(y, s, t, b, True, want_bits=True)
    r = timeit({"pass": lambda: K.bn_act_fwd(y, s, t, b, True, want_bits=True),
                "plain": lambda: K.conv2d(act, w, stats=True),
                "lazy": lambda: K.conv2d(K.Lazy(y, s, t, True, b), w, stats=True, join_out=True, join_bits=True)})
    print("join  %-28s pass %6.3f + plain %6.3f = %6.3f   lazy %6.3f   gain %+6.3f ms" %
          (name, r["pass"], r["plain"], r["pass"] + r["plain"], r["lazy"], r["pass"] + r["plain"] - r["lazy"]), flush=True)


def wgrad_case(name, hw, cin, cout, k, stride=1):
    pad = k // 2
    y = torch.randn(B, hw, hw, cin, device="cuda").bfloat16()
    oh = (hw + 2 * pad - k) // stride + 1
    dy = torch.randn(B, oh, oh, cout, device="cuda").bfloat16()
    s, t = coeffs(cin)
    act = K.bn_act_fwd(y, s, t, None, True)
    r = timeit({"plain": lambda: K.conv2d_wgrad(act, dy, k, k, stride, pad, pad),
                "lazy": lambda: K.conv2d_wgrad(K.Lazy(y, s, t, True), dy, k, k, stride, pad, pad)})
    print("wgrad %-28s plain %6.3f   lazy %6.3f   cost %+6.3f ms (a BatchNorm pass in the backward would cost a read + a write of x)" %
          (name, r["plain"], r["lazy"], r["lazy"] - r["plain"]), flush=True)


if __name__ == "__main__":
    print("B = %d" % B)
    if len(sys.argv) > 2 and sys.argv[2] == "wgrad":
        wgrad_case("l2.0 conv2 128->128 k3 s2 @224", 224, 128, 128, 3, 2)
        wgrad_case("l1.0 conv1 64->64 k1 @224 (stem)", 224, 64, 64, 1)
        wgrad_case("l1.0 downsample 64->256 k1 @224 (stem)", 224, 64, 256, 1)
        sys.exit(0)
    fwd_case("stem->l1 64->64 k1 @224", 224, 64, 64, 1)
    fwd_case("l1 conv2 64->64 k3 @224", 224, 64, 64, 3)
    fwd_case("l1 conv3 64->256 k1 @224", 224, 64, 256, 1)
    fwd_case("l2 conv2 128->128 k3 s2 @224", 224, 128, 128, 3, 2)
    fwd_case("l2 conv2 128->128 k3 @112", 112, 128, 128, 3)
    fwd_case("l2 conv3 128->512 k1 @112", 112, 128, 512, 1)
    fwd_case("l3 conv2 256->256 k3 @56", 56, 256, 256, 3)
    fwd_case("l3 conv3 256->1024 k1 @56", 56, 256, 1024, 1)
    fwd_case("l4 conv2 512->512 k3 @28", 28, 512, 512, 3)
    fwd_case("l4 conv3 512->2048 k1 @28", 28, 512, 2048, 1)
    join_case("l1 conv1 256->64 @224", 224, 256, 64)
    join_case("l2 conv1 512->128 @112", 112, 512, 128)
    join_case("l3 conv1 1024->256 @56", 56, 1024, 256)
    join_case("l4 conv1 2048->512 @28", 28, 2048, 512)
    wgrad_case("l2.0 conv2 128->128 k3 s2 @224", 224, 128, 128, 3, 2)
    wgrad_case("l1.0 conv1 64->64 k1 @224 (stem)", 224, 64, 64, 1)
    wgrad_case("l1 conv2 64->64 k3 @224", 224, 64, 64, 3)
    wgrad_case("l1 conv3 64->256 k1 @224", 224, 64, 256, 1)
    wgrad_case("l2 conv2 128->128 k3 @112", 112, 128, 128, 3)
    wgrad_case("l2 conv3 128->512 k1 @112", 112, 128, 512, 1)
    wgrad_case("l3 conv2 256->256 k3 @56", 56, 256, 256, 3)
    wgrad_case("l3 conv3 256->1024 k1 @56", 56, 256, 1024, 1)
