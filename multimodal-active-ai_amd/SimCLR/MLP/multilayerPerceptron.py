"""Drop-in for the reference's SimCLR/MLP/multilayerPerceptron.py on MI355X.

``MLP(input_dim, hidden_dim, output_dim)`` with ``.layers`` = Sequential(Linear,
ReLU, Linear) — state_dict keys ``layers.0.*`` / ``layers.2.*`` as in
multilayerPerceptron.py:9-16.  ``forward`` flattens in the caller's (NCHW)
order like ``x.view(B, -1)`` (``reshape``: the reference's ``view`` fails on the
channels-last tensor the 3-channel path produces) and runs both
GEMMs, bias and ReLU on the HIP engine.
"""
import os
import sys

import torch.nn as nn

try:
    import maai_hip  # noqa: F401
except ImportError:
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, "..", ".."), os.path.join(_h, "..", "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import engine as _engine


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, output_dim))

    def forward(self, x):
        return _engine.head_forward(self, x.reshape(x.size(0), -1))
