"""Eval-mode (frozen BatchNorm) backbone throughput: fused conv+BN epilogues vs conv + BN pass."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-active-ai_amd"); SIM = os.path.join(PKG, "SimCLR")
for d in (ROOT, PKG, SIM, os.path.join(SIM, "ResNet"), os.path.join(SIM, "MLP")):
    sys.path.insert(0, d)
import resnet as rn
from maai_hip import engine
B = int(os.environ.get("B", "256"))
f = rn.resnet50(crop_measures=1).cuda()
x = torch.randint(0, 256, (B, 3, 224, 224), device="cuda").float()
dtype = engine.compute_dtype()
f.train()
with torch.no_grad():
    engine.backbone_fwd(f, x, dtype, keep=False)
f.eval()
for flag in (1 << 40, 0, 250000, 1000000, 1 << 40, 0):   # rows up to which a unit uses the fused epilogue
    engine._EVAL_FUSE["max_rows"] = flag
    with torch.no_grad():
        for _ in range(2):
            engine.backbone_fwd(f, x, dtype, keep=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            engine.backbone_fwd(f, x, dtype, keep=False)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("eval forward, fused epilogues up to %d rows: %.1f ms  %.0f images/s" % (flag, dt * 1e3, B / dt), flush=True)
