import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
M, C = 256 * 112 * 112, 512   # 3.3 GB bf16 tensor
y = torch.randn(M, C, device="cuda").to(torch.bfloat16); r = torch.randn(M, C, device="cuda").to(torch.bfloat16)
sc = torch.rand(C, device="cuda"); sh = torch.rand(C, device="cuda"); out = torch.empty_like(y)
cfgs = [("0", ""), ("1", ""), ("2", ""), ("3", ""), ("4", ""), ("7", ""), ("0", "2048"), ("0", "16384"), ("4", "4096"), ("5", "4096")]
for res in (None, r):
    line = []
    for fl, gr in cfgs:
        os.environ["MAAI_EW_FLAGS"] = fl
        if gr: os.environ["MAAI_EW_GRID"] = gr
        else: os.environ.pop("MAAI_EW_GRID", None)
        best = 1e9
        for rnd in range(3):
            for _ in range(2): K.bn_act_fwd(y, sc, sh, res, True, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): K.bn_act_fwd(y, sc, sh, res, True, out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        nb = y.numel() * 2 * (3 if res is not None else 2)
        line.append("f%s%s: %.3fms %.0fGB/s" % (fl, ("/g" + gr) if gr else "", best, nb / best / 1e6))
    print(("res " if res is not None else "nores ") + " | ".join(line), flush=True)
