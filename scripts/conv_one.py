"""One conv shape, a few launches (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
from maai_hip import kernels as K
cin, cout, hw, k, B = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(cout, k, k, cin, device="cuda") / (cin * k * k) ** 0.5).to(torch.bfloat16)
for _ in range(3):
    K.conv2d(x, w, 1, k // 2, k // 2, stats=True)
torch.cuda.synchronize()
