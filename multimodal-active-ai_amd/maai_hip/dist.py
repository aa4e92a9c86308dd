"""Data-parallel glue over RCCL (torch.distributed backend 'nccl' on ROCm), one process per GPU.

Exchange steps of the SimCLR step at N > 1:

* embeddings — ``prefetch_embedding`` (called by ``SimCLR_Module.forward``): the L2-normalised projection of a view is
  all-gathered on a side HIP stream as soon as the forward that produced it has been enqueued, so view 1's gather
  (Objective.py:102-114, ``_cross_replica_concat``) runs under the whole view-2 forward; the loss picks the result up
  (``take_prefetched``) instead of gathering on the compute stream.
* gradients — the reference wraps the model in DDP and immediately unwraps it (Contrastive_Learning.py:418-424), so it
  never all-reduces gradients; a global-batch-4096 run needs that exchange.  ``GradReducer`` packs the
  gradients into a few flat buckets in the order the backward pass produces them (xGMI ring collectives are per-link
  bound: few large messages, not 161 small ones) and launches each bucket's all-reduce on the side stream THE MOMENT the
  engine's hand-written backward has produced its last gradient (``engine.set_grad_hook``) — the head's and layer4's
  buckets are in flight while layer3..1 are still being differentiated.  The averaged values are copied back into the
  gradient tensors autograd hands to ``.grad``: nothing aliases the buckets, so ``zero_grad(set_to_none=False)``,
  gradient accumulation and repeated calls are all safe.
* SyncBatchNorm statistics stay on the compute stream (the next kernel needs them): 2C doubles per layer, the two
  BatchNorms that meet at a projection shortcut share one collective (engine.py).

Every collective is issued in program order on the default process group — identical on every rank by construction."""
import weakref

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


_COMM = {}


def comm_stream(device):
    """The side HIP stream collectives are launched from (one per device); None for CPU tensors."""
    if device.type != "cuda":
        return None
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _COMM:
        _COMM[key] = torch.cuda.Stream(device=device)
    return _COMM[key]


# ----------------------------------------------------------------------------
# embedding all-gather, prefetched
# ----------------------------------------------------------------------------
_PREFETCH = {"key": None, "entry": None, "older": None}
STATS = {"prefetch_started": 0, "prefetch_hits": 0, "buckets_early": 0, "buckets_late": 0}   # diagnostics / tests


def _key(h):
    return (h.data_ptr(), tuple(h.shape), h.dtype, h.device)


def _alive(entry, h):
    """the prefetch belongs to ``h``'s storage as it is now: the tensor it was started from still exists, unmodified"""
    ref, version = entry[6], entry[7]
    t = ref()
    return t is not None and t._version == version and t.data_ptr() == h.data_ptr()


def prefetch_embedding(h, normalize_fn, world_size=None, group=None):
    """Start ``Z = all_gather(normalize(h))`` for a [B,d] embedding on the side stream and remember it under ``h``'s
    storage (``outputs1.data`` shares it, Contrastive_Learning.py:685).  ``normalize_fn(h) -> (z, inv_norm)``.  The two
    most recent prefetches are kept (hidden1 and hidden2 of one loss call)."""
    if not is_distributed():
        return
    world = dist.get_world_size(group) if world_size is None else world_size
    hd = h.detach()
    side = comm_stream(hd.device)
    if side is not None:
        cur = torch.cuda.current_stream(hd.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            z, inv = normalize_fn(hd.contiguous().float())
            out = torch.empty((world * z.shape[0], z.shape[1]), dtype=z.dtype, device=z.device)
            p2p = None
            if z.is_cuda:
                from . import comm
                p2p = comm.p2p_gather_for(z.numel() * 4, group)   # MAAI_P2P_GATHER=1: one-shot direct all-gather over xGMI
            if p2p is not None:
                p2p.gather(z.contiguous(), out)
                work = _Done(p2p)
                STATS["p2p_gathers"] = STATS.get("p2p_gathers", 0) + 1
            else:
                work = dist.all_gather_into_tensor(out, z, group=group, async_op=True)
        hd.record_stream(side)
    else:
        z, inv = normalize_fn(hd.contiguous().float())
        out = torch.empty((world * z.shape[0], z.shape[1]), dtype=z.dtype, device=z.device)
        work = dist.all_gather_into_tensor(out, z, group=group, async_op=True)
    STATS["prefetch_started"] += 1
    _PREFETCH["older"] = (_PREFETCH["key"], _PREFETCH["entry"])
    _PREFETCH["key"], _PREFETCH["entry"] = _key(hd), (work, z, inv, out, side, world, weakref.ref(h), h._version)


def take_prefetched(h, world_size):
    """(z, inv_norm, Z) of a prefetched embedding — the compute stream is made to wait for the gather — or None."""
    k = _key(h.detach())
    for slot in ("cur", "older"):
        key, entry = (_PREFETCH["key"], _PREFETCH["entry"]) if slot == "cur" else (_PREFETCH["older"] or (None, None))
        if entry is not None and key == k and entry[5] == world_size and _alive(entry, h):
            work, z, inv, out, side = entry[:5]
            STATS["prefetch_hits"] += 1
            work.wait()                         # orders the current stream behind the collective (direct transport: raises if a gather timed out)
            if side is not None:
                cur = torch.cuda.current_stream(z.device)
                cur.wait_stream(side)           # ... and behind the normalisation that fed it
                for t in (z, inv, out):
                    t.record_stream(cur)
            if slot == "cur":
                _PREFETCH["key"], _PREFETCH["entry"] = None, None
            else:
                _PREFETCH["older"] = None
            return z, inv, out
    return None


def drop_prefetched():
    for slot in ("entry",):
        e = _PREFETCH[slot]
        if e is not None:
            e[0].wait()
    o = _PREFETCH["older"]
    if o and o[1] is not None:
        o[1][0].wait()
    _PREFETCH["key"] = _PREFETCH["entry"] = _PREFETCH["older"] = None


# ----------------------------------------------------------------------------
# gradient averaging, overlapped with the backward pass
# ----------------------------------------------------------------------------
class GradReducer(object):
    """Bucketed gradient averaging.  Every bucket carries, behind the gradient data, one PRESENCE word per parameter (1.0
    where this rank produced a gradient): after the all-reduce it holds the number of ranks that did.  A parameter that
    got a gradient on some ranks only is then averaged and written back on EVERY rank (a rank without one receives a new
    tensor), so the replicas apply the same update; one that got none anywhere stays without (``None``)."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):  # backward order: the head and the last stage are ready first
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._next = 0          # first bucket not yet launched in the current backward
        self._inflight = []     # (work, flat, views, tensors, bucket)
        self.launched_early = 0  # buckets launched from inside the backward pass (diagnostics / tests)

    @staticmethod
    def _producer_streams(grads):
        """streams other than the compute stream that gradients in ``grads`` were produced on (the engine's weight-gradient
        side stream, MAAI_WGRAD_SIDE_STREAM=1): the communication stream has to wait for them too"""
        if grads is not None and grads.get("_side"):
            from . import engine
            return [engine._side_stream()]
        return []

    # -- called by the engine while the backward pass runs ---------------------------------------------------
    def ready(self, grads):
        """``grads``: {id(param): gradient tensor} filled so far.  Launches every leading bucket that is complete."""
        if not is_distributed():
            return
        while self._next < len(self.buckets) and all(id(p) in grads and grads[id(p)] is not None for p in self.buckets[self._next]):
            self._launch(self._next, [grads[id(p)] for p in self.buckets[self._next]], self._producer_streams(grads))
            self._next += 1
            self.launched_early += 1
            STATS["buckets_early"] += 1

    def finish(self, grads=None):
        """Launch what is left (a gradient missing on this rank travels as zeros with presence 0), wait for every bucket
        on the side stream, average, copy back; the compute stream continues behind the side stream.  Returns
        {id(param): tensor} of the gradients this rank did not have but other ranks did (also put into ``grads``)."""
        if not is_distributed():
            self._next = 0
            return {}
        while self._next < len(self.buckets):
            tensors = []
            for p in self.buckets[self._next]:
                g = None if grads is None else grads.get(id(p))
                tensors.append(g)
            self._launch(self._next, tensors, self._producer_streams(grads))
            self._next += 1
            STATS["buckets_late"] += 1
        world = dist.get_world_size(self.group)
        dev = self.params[0].device
        side = comm_stream(dev)
        ctx = torch.cuda.stream(side) if side is not None else _Null()
        created = {}
        with ctx:
            for work, flat, views, tensors, bucket in self._inflight:
                work.wait()
                n = sum(p.numel() for p in bucket)
                flat[:n].mul_(1.0 / world)
                dst = [t for t in tensors if t is not None]
                src = [v.view_as(t) for v, t in zip(views, tensors) if t is not None]
                if dst:
                    torch._foreach_copy_(dst, src)
                if len(dst) != len(tensors):
                    # (only a rank that is missing gradients reads the presence counts: one host read, off the common path)
                    counts = flat[n:n + len(bucket)].tolist()
                    for p, v, t, c in zip(bucket, views, tensors, counts):
                        if t is None and c > 0.5:
                            created[id(p)] = v.view_as(p).clone()
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            for t in created.values():
                t.record_stream(torch.cuda.current_stream(dev))
        if grads is not None:
            grads.update(created)
        self._inflight = []
        self._next = 0
        return created

    def _launch(self, i, tensors, producers=()):
        bucket = self.buckets[i]
        dev = self.params[0].device
        n = sum(p.numel() for p in bucket)
        total = n + len(bucket)   # gradient data | one presence word per parameter
        if self._flat[i] is None or self._flat[i].numel() != total or self._flat[i].device != dev:
            self._flat[i] = torch.empty(total, dtype=torch.float32, device=dev)
        flat = self._flat[i]
        views, off = [], 0
        for p in bucket:
            views.append(flat[off:off + p.numel()])
            off += p.numel()
        side = comm_stream(dev)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))   # the gradient kernels are enqueued on the compute stream
            for st in producers:
                side.wait_stream(st)                           # ... or on the engine's weight-gradient stream
        ctx = torch.cuda.stream(side) if side is not None else _Null()
        with ctx:
            have = [(v, t) for v, t in zip(views, tensors) if t is not None]
            if len(have) != len(views):
                flat[:n].zero_()
            if have:
                torch._foreach_copy_([v.view_as(t) for v, t in have], [t.detach() for _, t in have])
            if len(have) == len(views):
                flat[n:].fill_(1.0)
            else:
                flat[n:].copy_(torch.tensor([0.0 if t is None else 1.0 for t in tensors], dtype=torch.float32), non_blocking=False)
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        if side is not None:
            for t in tensors:
                if t is not None:
                    t.record_stream(side)
        self._inflight.append((work, flat, views, tensors, bucket))

    # -- the whole exchange after a backward that ran without the hook --------------------------------------
    def __call__(self):
        """Average ``.grad`` over the process group, in place (every bucket is launched now)."""
        if not is_distributed():
            return
        grads = {id(p): p.grad for p in self.params if p.grad is not None}
        self._next = 0
        created = self.finish(grads)
        for p in self.params:
            if id(p) in created:
                p.grad = created[id(p)]


class _Done(object):
    """stands in for the Work handle of a collective that was enqueued as an ordinary kernel on the side stream; with the
    direct transport ``wait`` also checks (without synchronising) that no finished gather gave up on a peer"""

    def __init__(self, p2p=None):
        self.p2p = p2p

    def wait(self):
        if self.p2p is not None:
            self.p2p.check()
        return True


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


GradAllReduce = GradReducer   # the round-1 name
