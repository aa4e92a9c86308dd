"""A/B of the 3x3 stride-1 weight-gradient kernels at the step's shapes (256 images): six-wave patch kernel / ping-pong /
eight-wave wide patch kernel (conv_wgrad3w.hip), interleaved, HIP events.  python3 scripts/wgrad3w_ab.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-active-ai_amd"))
import torch
from maai_hip import kernels as K
from maai_hip._lib import lib

SHAPES = [(256, 56, 56, 256, 256, 1), (256, 28, 28, 512, 512, 1), (256, 112, 112, 128, 128, 1), (256, 224, 224, 64, 64, 1),
          (256, 224, 224, 128, 128, 2), (256, 112, 112, 256, 256, 2), (256, 56, 56, 512, 512, 2)]
CONFIGS = [("patch", dict(MAAI_WGRAD_WIDE="0", MAAI_WGRAD_PP="0")), ("pingpong", dict(MAAI_WGRAD_WIDE="0", MAAI_WGRAD_PP="2")),
           ("wide16", dict(MAAI_WGRAD_WIDE="2", MAAI_WGRAD_WIDE_PW="16")), ("wide32", dict(MAAI_WGRAD_WIDE="2", MAAI_WGRAD_WIDE_PW="32"))]
TARGETS = [0, 256, 512, 768, 1536, 3072]


def run(x, dy, envs, target, stride=1):
    old = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    try:
        ms = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.conv2d_wgrad(x, dy, 3, 3, stride, 1, 1, target_blocks=target)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        name = lib().maai_last_kernel_name()
        return min(ms[2:]), (name or b"").decode()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    lib().maai_kernel_names(1)
    out = {}
    for n, h, w, cin, cout, stride in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, h, w, cin, generator=g).cuda().bfloat16()
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        dy = (torch.randn(n, oh, ow, cout, generator=g) * 0.05).cuda().bfloat16()
        flops = 2.0 * n * oh * ow * cout * 9 * cin
        row = {}
        for tag, envs in CONFIGS:
            best = None
            for t in TARGETS:
                if stride == 2 and tag == "wide32":
                    continue
                ms, name = run(x, dy, envs, t, stride)
                if best is None or ms < best[0]:
                    best = (ms, t, name)
            if best is None:
                continue
            row[tag] = {"ms": round(best[0], 4), "target": best[1], "PFLOPs": round(flops / best[0] / 1e12, 3), "kernel": best[2][:60]}
            # (the memset of dw is inside the timed call for every config alike)
        out["%dx%dx%dx%d->%d%s" % (n, h, w, cin, cout, " s2" if stride == 2 else "")] = row
        print(n, h, w, cin, cout, stride, json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
