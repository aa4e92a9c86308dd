"""Per-shape table of ONE backbone forward without a backward pass (training-mode BatchNorm, or eval with MODE=eval)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-active-ai_amd"); SIM = os.path.join(PKG, "SimCLR")
for d in (ROOT, PKG, SIM, os.path.join(SIM, "ResNet"), os.path.join(SIM, "MLP")):
    sys.path.insert(0, d)
import resnet as rn
from maai_hip import engine, kernels as K
B = int(os.environ.get("B", "256"))
f = rn.resnet50(crop_measures=1).cuda()
x = torch.randint(0, 256, (B, 3, 224, 224), device="cuda").float()
dtype = engine.compute_dtype()
f.train()
with torch.no_grad():
    for _ in range(2):
        engine.backbone_fwd(f, x, dtype, keep=False)
    if os.environ.get("MODE", "train") == "eval":
        f.eval()
        engine.backbone_fwd(f, x, dtype, keep=False)
    K.DETAIL[0] = True
    with K.profile() as prof:
        engine.backbone_fwd(f, x, dtype, keep=False)
    K.DETAIL[0] = False
t = prof.table() if hasattr(prof, "table") else prof
rows = sorted(t.items(), key=lambda kv: -kv[1]["ms"])
print("total %.1f ms" % sum(v["ms"] for v in t.values()))
for k, v in rows[:int(os.environ.get("TOP", "40"))]:
    ms = v["ms"]
    print("%-72s n%3d %6.2f ms %5.0f TF/s %5.0f GB/s" % (k[:72], v["launches"], ms, v["flops"] / ms / 1e9 if v["flops"] else 0, v["bytes"] / ms / 1e6))
