"""Host-side helpers the drivers import from ``Utilities`` (reference: SimCLR/Utilities.py:8-47): a running-mean
meter, the logging all-reduce (RCCL through torch.distributed's 'nccl' backend) and a scalar read-out.  Same names,
attributes and results as the reference; written independently."""
import torch.distributed as dist


class AverageMeter(object):
    """Tracks the latest sample (``val``), the weighted total (``sum``), the weight (``count``) and their ratio
    (``avg``) — the four attributes the training and validation loops read."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.sum = 0
        self.count = 0
        self.avg = 0

    def update(self, val, n=1):
        total = self.sum + val * n
        weight = self.count + n
        self.val, self.sum, self.count, self.avg = val, total, weight, total / weight


def reduce_tensor(tensor, world_size):
    """Mean of ``tensor`` over the ranks of the default group; the argument is left untouched."""
    mean = tensor.detach().clone() if tensor.requires_grad else tensor.clone()
    dist.all_reduce(mean, op=dist.ReduceOp.SUM)   # (the reference spells it dist.reduce_op, deprecated)
    return mean.div_(world_size)


def to_python_float(t):
    """Python scalar of a 0-d / 1-element tensor (or of the first entry of an indexable)."""
    item = getattr(t, "item", None)
    return item() if callable(item) else t[0]
