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
for fast, flag in ((False, 1 << 40), (False, 0), (True, 0), (True, 65536), (False, 0), (True, 65536)):
    # fast: units beyond ``flag`` rows still fuse where the epilogue runs on the plain launch's kernel (engine._EVAL_FUSE)
    engine._EVAL_FUSE["max_rows"] = flag
    engine._EVAL_FUSE["fast"] = fast
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
    print("eval forward, fused epilogues up to %d rows%s: %.1f ms  %.0f images/s" % (flag, " + fast-kernel rule" if fast else "", dt * 1e3, B / dt), flush=True)

if os.environ.get("DETAIL", "0") != "0":   # per-shape table of one eval forward at the default setting
    from maai_hip import kernels as K
    engine._EVAL_FUSE["max_rows"] = int(os.environ.get("MAAI_EVAL_FUSE_MAX_ROWS", "65536"))
    engine._EVAL_FUSE["fast"] = os.environ.get("MAAI_EVAL_FUSE_FAST", "1") != "0"
    K.DETAIL[0] = True
    with torch.no_grad(), K.profile() as prof:
        engine.backbone_fwd(f, x, dtype, keep=False)
    K.DETAIL[0] = False
    rows = sorted(prof.table().items(), key=lambda kv: -kv[1]["ms"])
    print("total %.1f ms over %d launches" % (sum(v["ms"] for _, v in rows), sum(v["launches"] for _, v in rows)))
    for k, v in rows:
        print("%-72s n %2d %6.2f ms %5.0f TF/s %5.0f GB/s" % (k, v["launches"], v["ms"], v["flops"] / v["ms"] / 1e9, v["bytes"] / v["ms"] / 1e6))
