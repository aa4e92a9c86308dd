"""Data-parallel glue over RCCL (torch.distributed backend 'nccl' on ROCm).

The reference wraps the model in DDP and immediately unwraps it
(Contrastive_Learning.py:418-424), so it never all-reduces gradients;
a global-batch-4096 run needs that exchange, so it is supplied here: gradients
are packed into a few large flat buckets (xGMI ring collectives are per-link
bound — few big messages, not 161 small ones), summed with one all_reduce per
bucket on a side stream, averaged with one kernel per bucket and handed back as views of the bucket."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradAllReduce(object):
    def __init__(self, params, bucket_bytes=128 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):  # backward order: last layers are ready first
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._stream = None

    def _comm_stream(self, device):
        if self._stream is None and device.type == "cuda":
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def __call__(self):
        """Average .grad over the process group (in place)."""
        if not is_distributed():
            return
        world = dist.get_world_size(self.group)
        works = []
        dev = self.params[0].device
        side = self._comm_stream(dev)
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            n = sum(g.numel() for g in grads)
            if self._flat[i] is None or self._flat[i].numel() != n:
                self._flat[i] = torch.empty(n, dtype=torch.float32, device=dev)
            flat = self._flat[i]
            torch.cat([g.reshape(-1) for g in grads], out=flat)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    works.append((dist.all_reduce(flat, group=self.group, async_op=True), flat, bucket))
            else:
                works.append((dist.all_reduce(flat, group=self.group, async_op=True), flat, bucket))
        for work, flat, bucket in works:
            work.wait()
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            # one division per bucket; the gradients become views of the flat buffer (161 per-tensor kernels otherwise,
            # each shorter than its launch).  The next backward replaces them before the buffer is packed again.
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
