"""Drop-in for the reference's SimCLR/Utilities.py (:8-47): host-side meters and
the logging all-reduce (RCCL through torch.distributed's 'nccl' backend)."""
import torch.distributed as dist


class AverageMeter(object):
    """running value / sum / count / mean"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def reduce_tensor(tensor, world_size):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= world_size
    return rt


def to_python_float(t):
    return t.item() if hasattr(t, 'item') else t[0]
