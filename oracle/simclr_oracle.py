"""CPU oracle for the SimCLR contrastive-pretraining hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product path (``multimodal-active-ai_amd/``) never imports this module and
fails loudly when the HIP library is missing.

What it is: a from-scratch *functional* restatement (plain fp32 torch-CPU
tensor ops over a flat ``state_dict``; no ``nn.Module``) of the reference's
algorithm for the path

    SimCLR/SimCLR.py:23-31          view packing + g(f(x))
    SimCLR/ResNet/resnet.py:31-243  BasicBlock / Bottleneck / ResNet._forward_impl
    torch BatchNorm2d semantics     (norm_layer, resnet.py:54,106-110,171)
    SimCLR/MLP/multilayerPerceptron.py:9-22
    SimCLR/Objective.py:17-81,123-125   NT-Xent (TF-SimCLR form)
    SimCLR/Model_Util.py:9-60,104-113   LR schedule, top-k accuracy
    SimCLR/SimCLR.py:36-144         legacy compute_loss
    Contrastive_Learning.py:638-700 train-step semantics (h1 detached, Adam)

Pinning: ``tests/golden/make_golden.py`` imports the reference's own modules
from /root/reference in the build container, runs them on seeded inputs with
closed-form weights (``pattern_state_dict`` below) and commits the outputs as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those files.  Parity is therefore PINNED to the reference itself.

``storage="bf16"`` additionally rounds tensors to bf16 at exactly the points
where the HIP path stores bf16 in HBM (weights, conv outputs, activations,
pooled features, hidden layer), keeping fp32 accumulation and fp32 BN
statistics, so that kernels can be compared at tight tolerance.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

LARGE_NUM = 1e9  # Objective.py:6
BN_EPS = 1e-5  # torch.nn.BatchNorm2d default used by resnet.py
BN_MOMENTUM = 0.1

# ----------------------------------------------------------------------------
# architecture tables (resnet.py:256-293)
# ----------------------------------------------------------------------------
ARCHS = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
    "resnet152": ("bottleneck", [3, 8, 36, 3]),
}


def expansion(arch: str) -> int:
    return 1 if ARCHS[arch][0] == "basic" else 4


def block_plan(arch: str) -> List[dict]:
    """Flat list of residual blocks with their conv shapes (resnet.py:201-224)."""
    kind, layers = ARCHS[arch]
    exp = expansion(arch)
    inplanes = 64
    plan = []
    for li, (planes, nblocks) in enumerate(zip([64, 128, 256, 512], layers)):
        for bi in range(nblocks):
            stride = 2 if (li > 0 and bi == 0) else 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            plan.append(dict(prefix=f"f.layer{li + 1}.{bi}", kind=kind, inplanes=inplanes,
                             planes=planes, stride=stride, downsample=has_ds))
            inplanes = planes * exp
    return plan


def backbone_param_shapes(arch: str, crop_measures: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of SimCLR_Module.f."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    shapes["f.conv1.weight"] = (64, 3 * crop_measures, 7, 7)
    bn("f.bn1", 64)
    exp = expansion(arch)
    for b in block_plan(arch):
        p, inp, pl = b["prefix"], b["inplanes"], b["planes"]
        if b["kind"] == "basic":
            shapes[p + ".conv1.weight"] = (pl, inp, 3, 3)
            bn(p + ".bn1", pl)
            shapes[p + ".conv2.weight"] = (pl, pl, 3, 3)
            bn(p + ".bn2", pl)
        else:
            shapes[p + ".conv1.weight"] = (pl, inp, 1, 1)
            bn(p + ".bn1", pl)
            shapes[p + ".conv2.weight"] = (pl, pl, 3, 3)
            bn(p + ".bn2", pl)
            shapes[p + ".conv3.weight"] = (pl * exp, pl, 1, 1)
            bn(p + ".bn3", pl * exp)
        if b["downsample"]:
            shapes[p + ".downsample.0.weight"] = (pl * exp, inp, 1, 1)
            bn(p + ".downsample.1", pl * exp)
    return shapes


def head_param_shapes(in_dim: int, hid: int = 1024, out: int = 128) -> Dict[str, Tuple[int, ...]]:
    return {"g.layers.0.weight": (hid, in_dim), "g.layers.0.bias": (hid,),
            "g.layers.2.weight": (out, hid), "g.layers.2.bias": (out,)}


def _hash_uniform(n: int, t: int) -> torch.Tensor:
    """Exact integer hash -> uniform(-1, 1), float64.  Pure int64 arithmetic
    (three multiplicative-congruential rounds mod 2^31-1), so the values are
    bit-identical on every platform; no RNG state involved."""
    m = 2147483647
    i = torch.arange(1, n + 1, dtype=torch.int64)
    x = (i * 48271 + (t + 1) * 69621) % m
    x = (x * 48271 + 12345) % m
    x = (x * 69621 + (t + 1) * 16807) % m
    x = (x * 48271) % m
    return (x.to(torch.float64) + 0.5) / m * 2.0 - 1.0


def pattern_state_dict(arch: str, crop_measures: int, head_in: int, hid: int = 1024,
                       out: int = 128, phase: int = 0, residual_gamma: float = 1.0) -> Dict[str, torch.Tensor]:
    """Closed-form weights (no RNG state): per-tensor integer-hash uniforms at
    kaiming scale.  conv weight: uniform with std sqrt(2/fan_out) (resnet.py:185
    kaiming_normal_(fan_out, relu) scale); linear weight/bias: uniform(+-1/sqrt(fan_in))
    (nn.Linear default scale); BN gamma = 1 + 0.2u, beta = 0.2u; running stats
    at their defaults.  ``residual_gamma`` scales gamma of the last BN of every
    residual branch (the knob resnet.py:194-199 ``zero_init_residual`` sets to 0):
    at 1.0 a randomly initialised ResNet-50 doubles any perturbation per block
    (fp32-vs-fp64 already differ by 3 % at the output), at 0.25 it is
    well-conditioned, which is what the ResNet-50 fixtures use."""
    shapes = dict(backbone_param_shapes(arch, crop_measures))
    last_bn = {b["prefix"] + (".bn2.weight" if b["kind"] == "basic" else ".bn3.weight") for b in block_plan(arch)}
    shapes.update(head_param_shapes(head_in, hid, out))
    sd: Dict[str, torch.Tensor] = {}
    for t, (name, shp) in enumerate(shapes.items()):
        n = int(math.prod(shp)) if len(shp) else 1
        tt = t + 1000 * phase
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shp)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shp)
        elif len(shp) == 4:  # conv
            std = math.sqrt(2.0 / (shp[0] * shp[2] * shp[3]))
            sd[name] = (std * math.sqrt(3.0) * _hash_uniform(n, tt)).float().reshape(shp)
        elif len(shp) == 2:  # linear
            sd[name] = (_hash_uniform(n, tt) / math.sqrt(shp[1])).float().reshape(shp)
        elif ".bn" in name or "downsample.1" in name:
            if name.endswith(".weight"):
                sd[name] = (1.0 + 0.2 * _hash_uniform(n, tt)).float().reshape(shp)
                if name in last_bn:
                    sd[name] = sd[name] * residual_gamma
            else:
                sd[name] = (0.2 * _hash_uniform(n, tt)).float().reshape(shp)
        else:  # linear bias: fan_in of the matching weight
            fan_in = shapes[name[:-4] + "weight"][1]
            sd[name] = (_hash_uniform(n, tt) / math.sqrt(fan_in)).float().reshape(shp)
    return sd


# ----------------------------------------------------------------------------
# rounding helper (bf16 storage emulation)
# ----------------------------------------------------------------------------
def _rnd(t: torch.Tensor, storage: str) -> torch.Tensor:
    if storage == "bf16":
        # straight-through rounding so autograd still works on the oracle
        return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()
    return t


# ----------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------
def pack_views(views: Sequence[torch.Tensor], batch_size: int, img_size: Tuple[int, int]) -> torch.Tensor:
    """SimCLR.py:24 — K x [B,H,W,3] u8 -> [B,3K,H,W] f32, channel = k*3+c, raw 0..255."""
    k = len(views)
    b, h, w, c = views[0].shape
    assert b == batch_size and (h, w) == tuple(img_size) and c == 3
    out = torch.empty(b, 3 * k, h, w, dtype=torch.float32)
    for vi, v in enumerate(views):
        for ci in range(3):
            out[:, vi * 3 + ci] = v[..., ci].to(torch.float32)
    return out


def batch_norm(y: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, training: bool,
               new_stats: Optional[dict], storage: str, y_stats: Optional[torch.Tensor] = None):
    """torch.nn.BatchNorm2d forward: biased var for normalisation, unbiased for
    running_var, momentum 0.1, eps 1e-5.  ``y_stats`` (fp32, un-rounded conv
    output) supplies the statistics in bf16-storage emulation."""
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        src = y if y_stats is None else y_stats
        m = src.shape[0] * src.shape[2] * src.shape[3]
        var, mean = torch.var_mean(src, dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            unbiased = var * (m / max(m - 1, 1))
            new_stats[prefix + ".running_mean"] = ((1 - BN_MOMENTUM) * sd[prefix + ".running_mean"]
                                                   + BN_MOMENTUM * mean).detach()
            new_stats[prefix + ".running_var"] = ((1 - BN_MOMENTUM) * sd[prefix + ".running_var"]
                                                  + BN_MOMENTUM * unbiased).detach()
            new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    invstd = torch.rsqrt(var + BN_EPS)
    return (y - mean[None, :, None, None]) * (invstd * g)[None, :, None, None] + b[None, :, None, None]


def _conv(x, w, stride, pad, storage):
    y32 = F.conv2d(x, _rnd(w, storage), None, stride=stride, padding=pad)
    return _rnd(y32, storage), y32


def stem_forward(sd, x, training=True, storage="fp32", new_stats=None):
    """resnet.py:228-230: conv1 (7x7, stride 1, pad 3) -> bn1 -> relu; max-pool NOT applied (:231)."""
    y, y32 = _conv(_rnd(x, storage), sd["f.conv1.weight"], 1, 3, storage)
    return _rnd(F.relu(batch_norm(y, sd, "f.bn1", training, new_stats, storage, y32)), storage)


def block_forward(sd, x, blk, training=True, storage="fp32", new_stats=None):
    """One residual block (BasicBlock resnet.py:59-77 / Bottleneck resnet.py:113-135, v1.5: stride on the 3x3)."""
    p = blk["prefix"]
    identity = x
    if blk["kind"] == "basic":
        y, y32 = _conv(x, sd[p + ".conv1.weight"], blk["stride"], 1, storage)
        o = _rnd(F.relu(batch_norm(y, sd, p + ".bn1", training, new_stats, storage, y32)), storage)
        y, y32 = _conv(o, sd[p + ".conv2.weight"], 1, 1, storage)
        o = batch_norm(y, sd, p + ".bn2", training, new_stats, storage, y32)
    else:
        y, y32 = _conv(x, sd[p + ".conv1.weight"], 1, 0, storage)
        o = _rnd(F.relu(batch_norm(y, sd, p + ".bn1", training, new_stats, storage, y32)), storage)
        y, y32 = _conv(o, sd[p + ".conv2.weight"], blk["stride"], 1, storage)
        o = _rnd(F.relu(batch_norm(y, sd, p + ".bn2", training, new_stats, storage, y32)), storage)
        y, y32 = _conv(o, sd[p + ".conv3.weight"], 1, 0, storage)
        o = batch_norm(y, sd, p + ".bn3", training, new_stats, storage, y32)
    if blk["downsample"]:
        y, y32 = _conv(x, sd[p + ".downsample.0.weight"], blk["stride"], 0, storage)
        identity = batch_norm(y, sd, p + ".downsample.1", training, new_stats, storage, y32)
    return _rnd(F.relu(o + identity), storage)


def backbone_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, arch: str, training: bool = True,
                     storage: str = "fp32", new_stats: Optional[dict] = None,
                     taps: Optional[dict] = None) -> torch.Tensor:
    """resnet.py:226-240: conv1 -> bn1 -> relu -> layer1..4; no maxpool/avgpool/fc."""
    x = stem_forward(sd, x, training, storage, new_stats)
    if taps is not None:
        taps["stem"] = x
    for blk in block_plan(arch):
        x = block_forward(sd, x, blk, training, storage, new_stats)
        if taps is not None:
            taps[blk["prefix"]] = x
    return x


def head_forward(sd: Dict[str, torch.Tensor], feat: torch.Tensor, storage: str = "fp32",
                 pool: Optional[int] = None) -> torch.Tensor:
    """multilayerPerceptron.py:18-22: flatten NCHW -> Linear -> ReLU -> Linear.
    ``pool``: adaptive avg-pool to pool x pool first (resnet.py:181's commented
    variant; identity at the native 4x4 map)."""
    if pool is not None and (feat.shape[2] != pool or feat.shape[3] != pool):
        feat = _rnd(F.adaptive_avg_pool2d(feat, (pool, pool)), storage)
    v = feat.reshape(feat.shape[0], -1)
    h = F.linear(v, _rnd(sd["g.layers.0.weight"], storage), sd["g.layers.0.bias"])
    h = _rnd(F.relu(h), storage)
    # the last GEMM runs in exact fp32 on the HIP path (fp32 weights, fp32 output)
    return F.linear(h, sd["g.layers.2.weight"], sd["g.layers.2.bias"])


def simclr_forward(sd, x, arch, training=True, storage="fp32", new_stats=None, pool=None, taps=None):
    return head_forward(sd, backbone_forward(sd, x, arch, training, storage, new_stats, taps), storage, pool)


# ----------------------------------------------------------------------------
# NT-Xent (Objective.py:17-81)
# ----------------------------------------------------------------------------
def l2_normalize(h: torch.Tensor) -> torch.Tensor:
    """F.normalize(h, dim=1, p=2): x / max(||x||, 1e-12) (Objective.py:42-43)."""
    return h / h.norm(dim=1, keepdim=True).clamp_min(1e-12)


def nt_xent(h1: torch.Tensor, h2: torch.Tensor, temperature: float = 1.0, hidden_norm: bool = True,
            rank: int = 0, world_size: int = 1,
            gathered: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Literal restatement.  ``gathered`` = (H1, H2) [N,d] already-normalised
    constants standing in for the cross-replica concat (Objective.py:102-114,
    no autograd through the gather).  Returns (loss, logits_ab, labels)."""
    if hidden_norm:
        h1, h2 = l2_normalize(h1), l2_normalize(h2)
    assert h1.shape == h2.shape
    b = h1.shape[0]
    if world_size > 1:
        assert gathered is not None
        H1, H2 = gathered
        n = H1.shape[0]
        idx = torch.arange(b) + rank * b
    else:
        H1, H2 = h1, h2
        n = b
        idx = torch.arange(b)
    labels = F.one_hot(idx, 2 * n)
    masks = F.one_hot(idx, n)
    aa = h1 @ H1.t() / temperature - masks * LARGE_NUM
    bb = h2 @ H2.t() / temperature - masks * LARGE_NUM
    ab = h1 @ H2.t() / temperature
    ba = h2 @ H1.t() / temperature

    def sce(t, x):  # Objective.py:123-125
        return -(t * F.log_softmax(x, dim=1)).sum() / x.shape[0]

    loss = sce(labels, torch.cat([ab, aa], 1)) + sce(labels, torch.cat([ba, bb], 1))
    return loss, ab, labels


def nt_xent_grad_h2(h1, h2, temperature=1.0, hidden_norm=True, rank=0, world_size=1, gathered=None):
    """d loss / d h2 (raw, pre-normalisation) with h1 detached — the gradient
    the train loop uses (Contrastive_Learning.py:685: hidden1=outputs1.data)."""
    h2 = h2.detach().clone().requires_grad_(True)
    loss, _, _ = nt_xent(h1.detach(), h2, temperature, hidden_norm, rank, world_size, gathered)
    (g,) = torch.autograd.grad(loss, h2)
    return loss.detach(), g


# ----------------------------------------------------------------------------
# legacy loss (SimCLR.py:36-144), kept because it is part of the named API
# ----------------------------------------------------------------------------
def legacy_compute_loss(z1: torch.Tensor, z2: torch.Tensor, temperature: float) -> torch.Tensor:
    n = z1.shape[0]
    z = torch.stack([z2, z1], dim=1).reshape(2 * n, -1)  # z[2k]=z2[k], z[2k+1]=z1[k]  (SimCLR.py:63-66)
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(1e-8)  # nn.CosineSimilarity eps
    s = zn @ zn.t()
    e = torch.exp(s / temperature)
    denom = e.sum(dim=1) - torch.diagonal(e)  # excludes exp(s[i,i]/t)  (SimCLR.py:43-46)
    total = z.new_zeros(())
    for k in range(n):
        i, j = 2 * k + 1, 2 * k
        total = total + (-torch.log(e[i, j] / denom[i])) + (-torch.log(e[j, i] / denom[j]))
    return total / 2 * n  # precedence quirk preserved (SimCLR.py:144)


# ----------------------------------------------------------------------------
# Model_Util restatements
# ----------------------------------------------------------------------------
def lr_at_step(global_step: int, base_lr: float, warmup_epochs: float, num_examples: int, batch_size: int,
               world_size: int, train_epochs: int, scaling: str = "linear") -> float:
    """Model_Util.py:9-60."""
    warmup_steps = int(round(warmup_epochs * num_examples // batch_size))
    gb = world_size * batch_size
    if scaling == "linear":
        scaled = base_lr * gb / 256.0
    elif scaling == "sqrt":
        scaled = base_lr * math.sqrt(gb)
    else:
        raise ValueError("Unknown learning rate scaling {}".format(scaling))
    lr = float(global_step) / int(warmup_steps) * scaled if warmup_steps else scaled
    total = num_examples * train_epochs // batch_size + 1
    if global_step >= warmup_steps:
        s, d = global_step - warmup_steps, total - warmup_steps
        s = min(s, d)
        lr = scaled * 0.5 * (1 + math.cos(math.pi * s / d))
    return lr


def top_k_accuracy(preds: torch.Tensor, target: torch.Tensor, k: int) -> torch.Tensor:
    """Model_Util.py:104-113."""
    top = torch.topk(preds, k=k, dim=1)[1]
    tgt = target if target.dim() == 1 else torch.argmax(target, dim=1)
    hit = (top == tgt[:, None]).any(dim=1)
    return hit.sum() / (hit.shape[0] + 0.0)


def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor update;
    ``step`` is the 1-based step count AFTER increment.  Returns new (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# ----------------------------------------------------------------------------
# whole training step (Contrastive_Learning.py:638-700, --num-fixations 1)
# ----------------------------------------------------------------------------
TRAINABLE_SUFFIX = (".weight", ".bias")


def trainable_keys(sd) -> List[str]:
    return [k for k in sd if k.endswith(TRAINABLE_SUFFIX)]


def train_step(sd, opt, x1, x2, arch, temperature, lr, storage="fp32", pool=None, h1_prev=None):
    """One image batch: view-1 forward (no grad, train-mode BN so running stats
    move), view-2 forward, NT-Xent with h1 detached, backward, Adam.  Mutates
    ``sd``/``opt`` in place; returns dict(loss, z1, z2, grads)."""
    if h1_prev is None:
        with torch.no_grad():
            ns = {}
            h1 = simclr_forward(sd, x1, arch, True, storage, ns, pool)
            sd.update(ns)
    else:
        h1 = h1_prev
    keys = trainable_keys(sd)
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaf)
    ns = {}
    h2 = simclr_forward(work, x2, arch, True, storage, ns, pool)
    loss, logits, labels = nt_xent(h1.detach(), h2, temperature)
    grads = torch.autograd.grad(loss, [leaf[k] for k in keys])
    sd.update(ns)
    opt["step"] = opt.get("step", 0) + 1
    for k, g in zip(keys, grads):
        m = opt.setdefault("m", {}).get(k, torch.zeros_like(g))
        v = opt.setdefault("v", {}).get(k, torch.zeros_like(g))
        p, m, v = adam_update(sd[k], g, m, v, opt["step"], lr)
        sd[k], opt["m"][k], opt["v"][k] = p.detach(), m, v
    return dict(loss=loss.detach(), z1=h1.detach(), z2=h2.detach(), logits=logits.detach(),
                grads=dict(zip(keys, grads)))


# ----------------------------------------------------------------------------
# augmentation restatement (north_star's crop / flip / colour-jitter two-view
# stage that replaces NVIDIA_DALI_Pipelines.py:444-480).  Integer/affine maths
# written so that the HIP kernel can match it bit-for-bit on u8 output.
# ----------------------------------------------------------------------------
def augment_view(img: torch.Tensor, params: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """img [H,W,3] u8; params f32[16] = (x0, y0, cw, ch, flip, brightness, contrast, M[3][3] row major) in
    source-pixel units; nearest-neighbour resize of the crop window to out_hw, horizontal flip, then colour twist
        v = ((src - 128)*contrast + 128) * brightness ;  out = clamp(M v)
    (M = hue rotation and saturation in YIQ, Contrastive_Learning.py:622-630 / DALI ColorTwist) evaluated in fp32,
    one rounding per operation, products summed left to right, round-half-up to u8."""
    H, W, _ = img.shape
    oh, ow = out_hw
    x0, y0, cw, ch, flip, br, ct = [float(v) for v in params[:7]]
    M = params[7:16].to(torch.float32).reshape(3, 3)
    ys = torch.arange(oh, dtype=torch.float32)
    xs = torch.arange(ow, dtype=torch.float32)
    if flip >= 0.5:
        xs = (ow - 1) - xs
    sy = torch.floor(torch.tensor(y0, dtype=torch.float32) + (ys + 0.5) * (torch.tensor(ch, dtype=torch.float32) / oh)).clamp(0, H - 1).long()
    sx = torch.floor(torch.tensor(x0, dtype=torch.float32) + (xs + 0.5) * (torch.tensor(cw, dtype=torch.float32) / ow)).clamp(0, W - 1).long()
    v = img[sy][:, sx].to(torch.float32)  # [oh, ow, 3]
    br32, ct32 = (torch.tensor(t, dtype=torch.float32) for t in (br, ct))
    v = ((v - 128.0) * ct32 + 128.0) * br32
    out = torch.stack([(M[c, 0] * v[..., 0] + M[c, 1] * v[..., 1]) + M[c, 2] * v[..., 2] for c in range(3)], dim=-1)
    return torch.floor(out.clamp(0.0, 255.0) + 0.5).clamp(0, 255).to(torch.uint8)


def colour_matrix(hue_deg: float, saturation: float) -> torch.Tensor:
    """fp64 reference of the matrix maai_augment_params builds: YIQ2RGB * R(hue) * diag(1, s, s) * RGB2YIQ."""
    import math
    a = torch.tensor([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]], dtype=torch.float64)
    b = torch.tensor([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]], dtype=torch.float64)
    h = math.radians(hue_deg)
    rot = torch.tensor([[1, 0, 0], [0, math.cos(h), -math.sin(h)], [0, math.sin(h), math.cos(h)]], dtype=torch.float64)
    return b @ rot @ torch.diag(torch.tensor([1.0, saturation, saturation], dtype=torch.float64)) @ a


# ----------------------------------------------------------------------------
# foveated retinal processor (csrc/foveate.hip; replaces NVIDIA_DALI_Pipelines.py:444-480).  numpy float32,
# one rounding per operation in the kernel's order, so the HIP output can be compared to 1 LSB.  Parity with
# DALI's own filters is unpinned (DALI cannot be installed here); THIS function is the pinned contract.
# ----------------------------------------------------------------------------
def _fov_hash(a, b, c, d):
    import numpy as np
    m = np.uint64(0xFFFFFFFF)
    a, b, c, d = (np.asarray(t, dtype=np.uint64) for t in (a, b, c, d))
    h = (a * np.uint64(0x9E3779B1) + np.uint64(0x85EBCA6B)) & m
    for t in (b, c, d):
        h = (h ^ ((t + np.uint64(0x9E3779B9) + ((h << np.uint64(6)) & m) + (h >> np.uint64(2))) & m)) & m
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & m
    h ^= h >> np.uint64(16)
    return h


def foveate_views(images, params, out_size=30):
    """images [B,H,W,3] u8 (numpy / tensor), params [B,32] f32 -> list of 4 arrays [B,OS,OS,3] u8."""
    import numpy as np
    f = np.float32
    img_all = np.asarray(images)
    P = np.asarray(params, dtype=np.float32)
    B, H, W, _ = img_all.shape
    OS = out_size
    outs = [np.zeros((B, OS, OS, 3), dtype=np.uint8) for _ in range(4)]
    CAN = f(640.0)
    for n in range(B):
        p = P[n]
        img = img_all[n].astype(np.float32)
        for view, S in enumerate((f(400), f(240), f(100), f(30))):
            ns = 1 if view == 3 else (3 if view == 2 else 4)
            ax = np.floor(p[29] * (CAN - S) + f(0.5)).astype(np.float32)
            ay = np.floor(p[30] * (CAN - S) + f(0.5)).astype(np.float32)
            s = (S / f(OS)).astype(np.float32)
            oy, ox = np.meshgrid(np.arange(OS, dtype=np.float32), np.arange(OS, dtype=np.float32), indexing="ij")
            acc = np.zeros((OS, OS, 3), dtype=np.float32)
            for ky in range(ns):
                for kx in range(ns):
                    cx = (ax + (ox + (f(kx) + f(0.5)) / f(ns)) * s) - f(0.5)
                    cy = (ay + (oy + (f(ky) + f(0.5)) / f(ns)) * s) - f(0.5)
                    fx = ((CAN - f(1.0)) - cx) if p[8] >= 0.5 else cx
                    fy = cy
                    v = np.zeros((OS, OS, 3), dtype=np.float32)
                    masked = np.zeros((OS, OS), dtype=bool)
                    if p[9] > 0:
                        gx = ((p[13] * fx) - (p[14] * fy)) + p[11]
                        gy = ((p[14] * fx) + (p[13] * fy)) + p[12]
                        tile = p[10]
                        ux = gx - np.floor(gx / tile) * tile
                        uy = gy - np.floor(gy / tile) * tile
                        lim = p[9] * tile
                        masked = (ux < lim) & (uy < lim)
                    dx, dy = fx - f(319.5), fy - f(319.5)
                    rx = ((p[6] * dx) + (p[7] * dy)) + f(319.5)
                    ry = ((p[6] * dy) - (p[7] * dx)) + f(319.5)
                    inside = (rx >= 0) & (rx <= CAN - f(1)) & (ry >= 0) & (ry <= CAN - f(1)) & ~masked
                    sx = (p[2] + (rx + f(0.5)) * (p[4] / CAN)) - f(0.5)
                    sy = (p[3] + (ry + f(0.5)) * (p[5] / CAN)) - f(0.5)
                    sx = np.minimum(np.maximum(sx, f(0)), p[1] - f(1))
                    sy = np.minimum(np.maximum(sy, f(0)), p[0] - f(1))
                    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
                    x1 = np.where(x0 + 1 < int(p[1]), x0 + 1, x0)
                    y1 = np.where(y0 + 1 < int(p[0]), y0 + 1, y0)
                    axw, ayw = sx - x0.astype(np.float32), sy - y0.astype(np.float32)
                    for c in range(3):
                        a_, b_ = img[y0, x0, c], img[y0, x1, c]
                        d_, e_ = img[y1, x0, c], img[y1, x1, c]
                        top = a_ + axw * (b_ - a_)
                        bot = d_ + axw * (e_ - d_)
                        v[..., c] = np.where(inside, top + ayw * (bot - top), f(0))
                    if p[16] > 0 or p[15] != 0:
                        seed = np.uint64(int(p[17]))
                        ix = np.floor(fx).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
                        iy = np.floor(fy).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
                        pid = view * 16 + ky * 4 + kx
                        for c in range(3):
                            h = _fov_hash(seed, ix, iy, np.uint64(c + 4 * pid))
                            u = ((h & np.uint64(0xff)) + ((h >> np.uint64(8)) & np.uint64(0xff)) + ((h >> np.uint64(16)) & np.uint64(0xff)) + (h >> np.uint64(24))).astype(np.float32)
                            z = (u - f(510.0)) * f(0.0067929)
                            v[..., c] = v[..., c] + (p[15] + p[16] * z)
                    for c in range(3):
                        m_ = ((p[18 + 3 * c] * v[..., 0]) + (p[19 + 3 * c] * v[..., 1])) + (p[20 + 3 * c] * v[..., 2])
                        acc[..., c] = acc[..., c] + ((p[27] * m_) + p[28])
            inv = (f(1.0) / f(ns * ns)).astype(np.float32)
            r = np.minimum(np.maximum(acc * inv, f(0)), f(255))
            outs[view][n] = np.floor(r + f(0.5)).astype(np.uint8)
    return outs
