"""Drop-in for the reference's SimCLR/Objective.py on MI355X: NT-Xent
(contrastive_loss, Objective.py:17-81) as fused HIP kernels — L2 normalise,
the 2N x 2N similarity on the fp32 matrix cores, -1e9 self mask, online
log-sum-exp, soft-label cross entropy and its backward — with one RCCL
all-gather of the normalised [h1|h2] when world_size > 1."""
import os
import sys

try:
    import maai_hip  # noqa: F401
except ImportError:
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, ".."), os.path.join(_h, "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import loss as _loss

LARGE_NUM = 1e9


def contrastive_loss(hidden1, hidden2, hidden_norm=True, temperature=1.0, local_rank=0, world_size=1, device='cpu'):
    """(loss, logits_ab, labels) — see maai_hip.loss.contrastive_loss."""
    return _loss.contrastive_loss(hidden1, hidden2, hidden_norm, temperature, local_rank, world_size, device)


def _cross_replica_concat(tensor, world_size, minibatch_size, dimensionality, device):
    """Objective.py:102-114: concatenation of ``tensor`` over ranks (no autograd)."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world_size * minibatch_size, dimensionality), dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(out, tensor.contiguous())
    return out


def _softmax_cross_entropy(targets, inputs):
    """Objective.py:123-125 (kept for API completeness; the loss above never materialises ``inputs``)."""
    import torch.nn.functional as f
    return -(targets * f.log_softmax(inputs, dim=1)).sum() / inputs.shape[0]
