"""Drop-in for the reference's SimCLR/SimCLR.py on MI355X.

``SimCLR_Module(f, g, batch_size, img_size, device)`` keeps the public
attributes ``f g batch_size img_size device`` (other code swaps ``g`` for
Identity and reads ``f``, Representation_Evaluation.py:415).  ``forward`` takes
the reference's input — a list of K uint8 HWC views ``[B,H,W,3]`` — and returns
``g(f(x))`` with x = the K views stacked on channels (k*3+c), raw 0..255 floats
(SimCLR.py:24).  When f/g are this package's ResNet/MLP the packing, backbone,
head and their backward run as one fused HIP pipeline in NHWC.
"""
import os
import sys

import torch
import torch.nn as nn

try:
    import maai_hip  # noqa: F401
except ImportError:
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, ".."), os.path.join(_h, "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import engine as _engine
from maai_hip import loss as _loss


def _is_hip_backbone(m):
    return all(hasattr(m, a) for a in ("conv1", "bn1", "layer1", "layer4")) and hasattr(m, "_forward_impl")


def _is_hip_head(m):
    return hasattr(m, "layers") and len(getattr(m, "layers", ())) == 3 and isinstance(m.layers[0], nn.Linear)


class SimCLR_Module(nn.Module):
    def __init__(self, f, g, batch_size, img_size, device):
        super().__init__()
        self.f, self.g = f, g
        self.batch_size, self.img_size, self.device = batch_size, img_size, device
        # adaptive pooling of the layer4 map before g (resnet.py:181's variant); identity at the native 4x4 map
        self.head_pool = None

    def forward(self, inputs):
        views = list(inputs)
        b = views[0].shape[0]
        if b != self.batch_size or tuple(views[0].shape[1:3]) != tuple(self.img_size):
            raise RuntimeError("SimCLR_Module: expected %d views of [%d,%d,%d,3], got %s" %
                               (len(views), self.batch_size, self.img_size[0], self.img_size[1], tuple(views[0].shape)))
        if _is_hip_backbone(self.f) and _is_hip_head(self.g) and views[0].dtype == torch.uint8:
            out = _engine.fused_forward(self.f, self.g, [v.contiguous() for v in views], self.head_pool)
            if self.training:
                # N > 1: the cross-replica gather of this view's embedding (Objective.py:102-114) starts now, on a
                # side stream — view 1's then runs under the whole view-2 forward (no-op at world size 1)
                _loss.prefetch_embedding(out)
            return out
        x = torch.stack(views).permute(1, 0, 4, 2, 3).reshape(self.batch_size, -1, self.img_size[0], self.img_size[1]).float()
        return self.g(self.f(x.contiguous()))

    def forward_tensor(self, x):
        """g(f(x)) for an already packed NCHW fp32 batch (BASELINE cfg1 feeds randn(64,3,32,32))."""
        if _is_hip_backbone(self.f) and _is_hip_head(self.g):
            out = _engine.fused_forward(self.f, self.g, x, self.head_pool)
            if self.training:
                _loss.prefetch_embedding(out)
            return out
        return self.g(self.f(x))


def compute_loss(z1, z2, temperature):
    """Legacy Algorithm-1 loss (SimCLR.py:36-144).  Each l(i,j) is -log softmax over
    the 2N-1 other cosine similarities, i.e. exactly one row of the NT-Xent above,
    so Sum = N * contrastive_loss and the reference's ``Sum / 2*N`` (precedence
    quirk kept, SimCLR.py:144) equals loss * N*N / 2."""
    n = z1.shape[0]
    loss, _, _ = _loss.contrastive_loss(z1, z2, True, temperature)
    return loss * n / 2 * n
