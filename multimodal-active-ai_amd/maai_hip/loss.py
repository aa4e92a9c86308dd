"""NT-Xent on the HIP kernels with the reference's autograd semantics
(Objective.py:17-81): gradients flow to hidden1/hidden2 through the local
operands; cross-replica copies are constants (Objective.py:112-114 gathers
without autograd), which at world_size 1 degenerates to the fully
differentiable single-process loss."""
import torch
import torch.distributed as dist

from . import dist as D
from . import kernels as K
from ._lib import MaaiError


def gather_normalized(z1, z2, world_size, group=None):
    """One all-gather of [z1 | z2] ([B,2d], 8*B*d bytes) instead of the reference's
    two list-based all_gathers + cat (Objective.py:102-114).  Returns (Z1, Z2) [N,d]."""
    b, d = z1.shape
    packed = torch.cat([z1, z2], dim=1).contiguous()
    out = torch.empty((world_size * b, 2 * d), dtype=z1.dtype, device=z1.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    return out[:, :d].contiguous(), out[:, d:].contiguous()


def gather_one(z, world_size, group=None):
    out = torch.empty((world_size * z.shape[0], z.shape[1]), dtype=z.dtype, device=z.device)
    dist.all_gather_into_tensor(out, z.contiguous(), group=group)
    return out


def prefetch_embedding(h, _normalize=None):
    """Start the all-gather of normalize(h) on the side stream (maai_hip.dist.prefetch_embedding); a later
    contrastive_loss(h.data, ...) / contrastive_loss(..., h) with hidden_norm=True picks it up.
    (``_normalize``: tests substitute the normalisation kernel, e.g. to inject a failure on one rank.)"""
    if _normalize is not None:
        if D.is_distributed() and h.dim() == 2:
            D.prefetch_embedding(h, _normalize)
        return
    if D.is_distributed() and h.is_cuda and h.dim() == 2:
        # Which collectives a step issues must be the same on every rank.  A failure here (say an out-of-memory error
        # in the normalisation) is therefore NOT absorbed by a rank-local fallback to the in-loss gather — the other
        # ranks would already have enqueued this [B,d] all-gather and the sequences would no longer pair up: the
        # exception propagates and the job ends non-zero.
        D.prefetch_embedding(h, lambda t: K.ntxent_normalize(t, True))


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h1, h2, hidden_norm, temperature, rank, world_size):
        if not h1.is_cuda:
            raise MaaiError("contrastive_loss: the HIP path needs HIP tensors (got %s); there is no CPU fallback" % h1.device)
        h1c, h2c = h1.contiguous().float(), h2.contiguous().float()
        b = h1c.shape[0]
        # gathers that SimCLR_Module.forward started on the side stream (view 1's ran under the view-2 forward)
        pre1 = D.take_prefetched(h1, world_size) if (world_size > 1 and hidden_norm) else None
        pre2 = D.take_prefetched(h2, world_size) if (world_size > 1 and hidden_norm) else None
        z1, inv1 = (pre1[0], pre1[1]) if pre1 is not None else K.ntxent_normalize(h1c, hidden_norm)
        z2, inv2 = (pre2[0], pre2[1]) if pre2 is not None else K.ntxent_normalize(h2c, hidden_norm)
        if world_size > 1:
            if pre1 is None and pre2 is None:
                Z1, Z2 = gather_normalized(z1, z2, world_size)
            else:
                Z1 = pre1[2] if pre1 is not None else gather_one(z1, world_size)
                Z2 = pre2[2] if pre2 is not None else gather_one(z2, world_size)
            off = rank * b
        else:
            Z1, Z2, off = z1, z2, 0
        loss, logits, lse = K.ntxent_fwd(z1, z2, Z1, Z2, temperature, off)
        ctx.save_for_backward(z1, z2, Z1, Z2, lse, inv1, inv2)
        ctx.cfg = (hidden_norm, temperature, off, world_size == 1)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        z1, z2, Z1, Z2, lse, inv1, inv2 = ctx.saved_tensors
        hidden_norm, temperature, off, local = ctx.cfg
        need1 = ctx.needs_input_grad[0]
        gl = gloss.contiguous().float()
        dz1, dz2 = K.ntxent_bwd(z1, z2, Z1, Z2, lse, gl, temperature, off, local, need1)
        dh2 = K.ntxent_normalize_bwd(z2, dz2, inv2, hidden_norm) if ctx.needs_input_grad[1] else None
        dh1 = K.ntxent_normalize_bwd(z1, dz1, inv1, hidden_norm) if need1 else None
        return dh1, dh2, None, None, None, None


def contrastive_loss(hidden1, hidden2, hidden_norm=True, temperature=1.0, local_rank=0, world_size=1, device='cpu'):
    """Reference signature and returns (Objective.py:17-81): (loss, logits_ab [B,N],
    labels [B,2N] int64 one-hot).  ``local_rank`` is the GLOBAL rank, as the driver
    passes it (Contrastive_Learning.py:688)."""
    from . import engine
    engine.flush_overlap()   # (a no-grad forward still in flight on the side stream: its output is read from here on)
    assert hidden1.shape == hidden2.shape
    loss, logits = _NTXentFn.apply(hidden1, hidden2, bool(hidden_norm), float(temperature), int(local_rank), int(world_size))
    b = hidden1.shape[0]
    n = logits.shape[1]
    idx = torch.arange(b, device=hidden1.device) + (local_rank * b if world_size > 1 else 0)
    labels = torch.nn.functional.one_hot(idx, 2 * n)
    return loss, logits, labels
