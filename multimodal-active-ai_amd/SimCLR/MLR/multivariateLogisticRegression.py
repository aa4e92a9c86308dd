"""Drop-in for the reference's SimCLR/MLR/multivariateLogisticRegression.py (:6-13): the linear-probe classifier of
Representation_Evaluation.py:598-712.  ``self.linear`` stays an ``nn.Linear`` (same state_dict keys); its forward and
backward run on the HIP kernels (maai_hip.probe: implicit-GEMM logits in exact fp32) whenever the features live on a
HIP device.  ``HipCrossEntropyLoss`` is the matching criterion on the library's softmax-CE kernel; importing this module
also makes the driver's own ``nn.CrossEntropyLoss()`` (Representation_Evaluation.py:455) dispatch to it for the default
configuration on HIP logits (maai_hip.probe.install_cross_entropy; MAAI_PATCH_CE=0 leaves torch.nn alone)."""
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
from maai_hip import probe as _probe


class LogisticRegression(nn.Module):
    def __init__(self, input_size, num_classes):
        super().__init__()
        self.linear = nn.Linear(input_size, num_classes)

    def forward(self, x):
        return _probe.linear(x, self.linear.weight, self.linear.bias)


HipCrossEntropyLoss = _probe.CrossEntropyLoss
if os.environ.get("MAAI_PATCH_CE", "1") != "0":
    _probe.install_cross_entropy(True)

