"""Optimisers of the path (Model_Util.py:68-88) on the fused HIP update kernels.
torch.optim.Optimizer subclasses so that param_groups / state / state_dict /
``optimizer.state[p]['step']`` (read by learning_rate_schedule, Model_Util.py:11-15)
behave like torch.optim.Adam / SGD."""
import torch

from . import kernels as K
from ._lib import MaaiError
from . import engine
from .engine import bump_weight_epoch


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics (no weight decay / amsgrad: the reference uses neither)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._multi = {}   # per group: (key, kernels.AdamMulti) — one launch for all tensors of the group

    @torch.no_grad()
    def step(self, closure=None):
        engine.flush_overlap()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            live = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                live.append(p)
            if not live:
                continue
            steps = {self.state[p]["step"] for p in live}
            plain = all(p.dtype == torch.float32 and p.is_contiguous() for p in live)
            if len(steps) == 1 and plain and len(live) > 1:
                # every tensor of the group in one launch (the per-tensor launches are shorter than their issue time)
                key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in live)
                hit = self._multi.get(gi)
                if hit is None or hit[0] != key:
                    hit = (key, K.AdamMulti(live, [self.state[p]["exp_avg"] for p in live], [self.state[p]["exp_avg_sq"] for p in live]))
                    self._multi[gi] = hit
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
                hit[1].step(grads, group["lr"], b1, b2, group["eps"], steps.pop())
            else:
                for p in live:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    K.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"], st["step"])
        bump_weight_epoch()
        return loss


class HipSGD(torch.optim.Optimizer):
    """torch.optim.SGD(params, lr, momentum, weight_decay) semantics (dampening 0, no nesterov)."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._multi = {}   # per group: (key, kernels.SgdMulti) — one launch for all tensors of the group

    @torch.no_grad()
    def step(self, closure=None):
        engine.flush_overlap()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            live, firsts = [], set()
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.zeros_like(p)
                    st["step"] = 0
                st["step"] += 1
                live.append(p)
                firsts.add(first)
            if not live:
                continue
            plain = all(p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda for p in live)
            if plain and len(firsts) == 1 and len(live) > 1:
                # every tensor of the group in one launch (maai_sgd_step_multi: same arithmetic as the per-tensor kernel)
                key = tuple((p.data_ptr(), self.state[p]["momentum_buffer"].data_ptr()) for p in live)
                hit = self._multi.get(gi)
                if hit is None or hit[0] != key:
                    hit = (key, K.SgdMulti(live, [self.state[p]["momentum_buffer"] for p in live]))
                    self._multi[gi] = hit
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
                hit[1].step(grads, group["lr"], group["momentum"], group["weight_decay"], firsts.pop())
            else:
                for p in live:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    K.sgd_step(p, g, st["momentum_buffer"], group["lr"], group["momentum"], group["weight_decay"],
                               st["step"] == 1)
        bump_weight_epoch()
        return loss


class LARC(object):
    """Layer-wise adaptive rate clipping around another optimiser, following
    Apex's published algorithm (apex/parallel/LARC.py: trust_coefficient 0.02,
    clip=True, eps 1e-8): each gradient is scaled by
    min(trust*||w||/(||g|| + wd*||w|| + eps) / lr, 1) before the wrapped step.
    Apex is not installable offline, so parity of this class is UNPINNED
    (SURVEY §8c); it exists so that ``--optimizer lars`` (Model_Util.py:80-83,
    = LARC(Adam)) keeps working."""

    def __init__(self, optimizer, trust_coefficient=0.02, clip=True, eps=1e-8):
        self.optim, self.trust_coefficient, self.clip, self.eps = optimizer, trust_coefficient, clip, eps

    def __getattr__(self, name):
        return getattr(self.__dict__["optim"], name)

    @property
    def state(self):
        return self.optim.state

    @property
    def param_groups(self):
        return self.optim.param_groups

    def state_dict(self):
        return self.optim.state_dict()

    def load_state_dict(self, sd):
        self.optim.load_state_dict(sd)

    def zero_grad(self, *a, **k):
        self.optim.zero_grad(*a, **k)

    def step(self):
        """Two launches per parameter group (all norms; all rescalings) around the wrapped optimiser's step; the
        group's own weight decay is folded into the gradient here and switched off for the inner step, as Apex does."""
        with torch.no_grad():
            saved = []
            for gi, group in enumerate(self.optim.param_groups):
                wd = group.get("weight_decay", 0.0)
                saved.append(wd)
                group["weight_decay"] = 0.0
                ps = [p for p in group["params"] if p.grad is not None]
                if not ps:
                    continue
                if not all(p.is_cuda for p in ps):
                    raise MaaiError("LARC runs on the HIP kernels: parameters must live on a HIP device (no CPU fallback)")
                key = tuple(id(p) for p in ps)
                cache = self.__dict__.setdefault("_multi", {})
                if cache.get(gi, (None, None))[0] != key:
                    cache[gi] = (key, K.LarcMulti(ps))
                multi = cache[gi][1]
                grads = []
                for p in ps:
                    if not p.grad.is_contiguous():
                        p.grad = p.grad.contiguous()
                    grads.append(p.grad)
                multi.sqnorms(grads)
                multi.scale(self.trust_coefficient, group["lr"], wd, self.eps, self.clip)
        self.optim.step()
        for group, wd in zip(self.optim.param_groups, saved):
            group["weight_decay"] = wd
