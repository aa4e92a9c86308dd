"""Linear probe on frozen features (Representation_Evaluation.py:598-712): the classifier is
``LogisticRegression`` = one ``nn.Linear(D*F, classes)`` (SimCLR/MLR/multivariateLogisticRegression.py:6-13) trained
with ``nn.CrossEntropyLoss``.  Both run on this library: the logits and their two gradients are implicit-GEMM launches
in exact fp32 (the class count is padded to the GEMM's 64-column granularity with zero rows), the loss is one
wave-per-row softmax cross-entropy kernel.  No fallback: HIP tensors only."""
import torch

from . import kernels as K
from ._lib import MaaiError


def _pad64(n):
    return (n + 63) // 64 * 64


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b for x [B, I] fp32, W [O, I], b [O]; returns [B, O]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not (x.is_cuda and weight.is_cuda):
            raise MaaiError("probe linear: the HIP path needs HIP tensors (got %s / %s); there is no CPU fallback" % (x.device, weight.device))
        b, i = x.shape
        o = weight.shape[0]
        if i % 16:
            raise MaaiError("probe linear: in_features must be a multiple of 16 (got %d)" % i)
        op = _pad64(o)
        x4 = x.detach().contiguous().float().reshape(b, 1, 1, i)
        w4 = torch.zeros((op, 1, 1, i), dtype=torch.float32, device=x.device)
        w4[:o, 0, 0] = weight.detach().float()
        y = K.conv2d(x4, w4)                                    # [B,1,1,op], exact-fp32 MFMA
        if bias is not None:
            bp = torch.zeros(op, dtype=torch.float32, device=x.device)
            bp[:o] = bias.detach().float()
            y = K.bn_act_fwd(y, None, bp, None, False)
        ctx.save_for_backward(x4, w4)
        ctx.o, ctx.has_bias = o, bias is not None
        return y.reshape(b, op)[:, :o]

    @staticmethod
    def backward(ctx, dy):
        x4, w4 = ctx.saved_tensors
        b, i = x4.shape[0], x4.shape[3]
        op, o = w4.shape[0], ctx.o
        d4 = torch.zeros((b, 1, 1, op), dtype=torch.float32, device=dy.device)
        d4[:, 0, 0, :o] = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d(d4, w4.reshape(op, i).t().contiguous().reshape(i, 1, 1, op)).reshape(b, i)
        if ctx.needs_input_grad[1]:
            dw = K.conv2d_wgrad(x4, d4, 1, 1).reshape(op, i)[:o]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = K.bn_act_bwd_reduce(d4, None, None, None, False)[:op].float()[:o]
        return dx, dw, db


def linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        if not logits.is_cuda:
            raise MaaiError("cross_entropy: the HIP path needs HIP tensors (got %s); there is no CPU fallback" % logits.device)
        lg = logits.contiguous().float()
        lb = labels.contiguous().to(torch.int64)
        loss, lse = K.softmax_ce_fwd(lg, lb, lg.shape[1])
        ctx.save_for_backward(lg, lb, lse)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        lg, lb, lse = ctx.saved_tensors
        return K.softmax_ce_bwd(lg, lb, lse, gloss.contiguous().float().reshape(1), lg.shape[1]), None


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss()(logits, labels): mean over the batch, class-index targets."""
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise MaaiError("cross_entropy: logits [B,C] and class-index labels [B]")
    return _CrossEntropyFn.apply(logits, labels)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the driver's ``criterion = nn.CrossEntropyLoss()`` (Representation_Evaluation.py:455) on the HIP kernel."""

    def forward(self, logits, labels):
        return cross_entropy(logits, labels)


_TORCH_CE = torch.nn.CrossEntropyLoss


class _HipAwareCrossEntropyLoss(_TORCH_CE):
    """``nn.CrossEntropyLoss`` as the UNCHANGED probe driver constructs it (Representation_Evaluation.py:455,
    ``criterion = nn.CrossEntropyLoss().to(device)``): the default configuration on HIP logits [B, C] with class-index targets [B]
    runs on the library's softmax-CE kernel; every other configuration (class weights, another reduction, label smoothing, a
    non-default ignore_index, probabilities as targets, tensors off the GPU) is torch's own forward, untouched."""

    def forward(self, input, target):
        if (input.is_cuda and input.dim() == 2 and target.dim() == 1 and target.dtype == torch.int64 and self.weight is None
                and self.reduction == "mean" and self.ignore_index == -100 and getattr(self, "label_smoothing", 0.0) == 0.0):
            return cross_entropy(input, target)
        return super().forward(input, target)


def install_cross_entropy(enable=True):
    """Make ``torch.nn.CrossEntropyLoss`` the HIP-aware subclass (or put torch's class back).  Called when the drop-in
    ``multivariateLogisticRegression`` module is imported, unless MAAI_PATCH_CE=0: the reference's probe driver then needs no edit."""
    cls = _HipAwareCrossEntropyLoss if enable else _TORCH_CE
    torch.nn.CrossEntropyLoss = cls
    torch.nn.modules.loss.CrossEntropyLoss = cls
    return cls

