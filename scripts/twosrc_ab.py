"""Timing of the folded units' two-source data-gradient launch at the step's layer-2 shape (256 images, 112^2, K = 512 + 128 -> 128)
under the ring kernel's tile-row choices:  python3 scripts/twosrc_ab.py   (MAAI_CONV_BM / MAAI_CONV_PP are read per call)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-active-ai_amd"))
import torch
from maai_hip import kernels as K
from maai_hip._lib import lib

lib().maai_kernel_names(1)
for (n, hw, c1, c2) in ((256, 112, 512, 128), (256, 56, 1024, 256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(n, hw, hw, c1, device="cuda", generator=g).bfloat16()
    b = torch.relu(torch.randn(n, hw, hw, c2, device="cuda", generator=g)).bfloat16()
    wcat = (torch.randn(c2, 1, 1, c1 + c2, device="cuda", generator=g) / (c1 + c2) ** 0.5).bfloat16()
    bias, mean, t2, dg = (torch.randn(c2, device="cuda", generator=g) * 0.2 for _ in range(4))
    s2 = torch.rand(c2, device="cuda", generator=g) + 0.5
    y2 = torch.randn(n, hw, hw, c2, device="cuda", generator=g).bfloat16()
    out = torch.empty_like(y2)
    for envs in ({}, {"MAAI_CONV_BM": "128"}, {"MAAI_CONV_BM": "256"}, {"MAAI_CONV_PP": "0"}, {"MAAI_CONV_PP": "0", "MAAI_CONV_BM": "128"}):
        old = {k: os.environ.get(k) for k in envs}
        os.environ.update(envs)
        try:
            rows = K.conv2d_stats_rows(a, wcat, 1, 0, 0, x2=b)
            slab = torch.empty((rows, 2, c2), dtype=torch.float32, device="cuda")
            ms = []
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.conv2d_store_reduce(a, wcat, 1, 0, 0, out, slab, y2, mean, s2, t2, None, x2=b, bias=bias, diag=dg)
                e1.record()
                e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            name = (lib().maai_last_kernel_name() or b"").decode()[:70]
            gb = 2e-9 * n * hw * hw * (c1 + 3 * c2)
            print("%dx%d^2 %d+%d->%d %-40s %.3f ms  %.2f TB/s  %s" % (n, hw, c1, c2, c2, envs, min(ms[2:]), gb / min(ms[2:]), name), flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
