#!/usr/bin/env python3
"""Probe (not a product path): how much would running the no-grad view-1 forward on a second HIP stream, concurrently with
the view-2 forward, buy?  The two forwards are independent until the loss.  Running statistics are then updated in an
undefined order (the probe ignores that; a product version would have to apply the two updates in order afterwards).
   python scripts/overlap_probe.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    gpus, batch, img, arch, temperature, recompute = 1, 256, 224, "resnet50", 0.5, False


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    model, opt = bench.build(A, dev, 1)
    import Objective
    from maai_hip import kernels as K, engine
    images = bench.synthetic_images(A.batch, A.img, dev, 1234)
    side = torch.cuda.Stream()
    dtype = engine.compute_dtype()

    def prewarm():
        # every weight-derived cache both forwards read, rebuilt on the main stream before the fork
        for m in model.f.modules():
            if isinstance(m, torch.nn.Conv2d) and m is not model.f.conv1:
                engine.w_fwd(m.weight, dtype)
        engine.w_stem_unrolled(model.f.conv1.weight, dtype)
        l0, l2 = model.g.layers[0], model.g.layers[2]
        engine.w_linear(l0.weight, dtype, (2048, 16))
        engine.w_linear(l2.weight, torch.float32)

    def step(overlap, it):
        p1 = K.augment_params(A.batch, A.img, A.img, seed=1000, view=2 * it, device=dev)
        p2 = K.augment_params(A.batch, A.img, A.img, seed=1000, view=2 * it + 1, device=dev)
        v1 = K.augment_view_u8(images, p1, A.img, A.img)
        v2 = K.augment_view_u8(images, p2, A.img, A.img)
        if overlap:
            prewarm()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.no_grad():
                    h1 = model([v1])
            h2 = model([v2])
            torch.cuda.current_stream().wait_stream(side)
        else:
            with torch.no_grad():
                h1 = model([v1])
            h2 = model([v2])
        loss, _, _ = Objective.contrastive_loss(hidden1=h1.data, hidden2=h2, temperature=0.5, device=dev)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for mode in (False, True, False, True):
        for i in range(2):
            step(mode, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(mode, 10 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("overlap=%s  %.1f ms/step  %.1f images/s  loss %.4f" % (mode, dt * 1e3, A.batch / dt, float(loss)), flush=True)


if __name__ == "__main__":
    main()
