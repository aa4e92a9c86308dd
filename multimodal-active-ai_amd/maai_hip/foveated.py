"""Host side of the foveated retinal processor (csrc/foveate.hip): turns the per-batch augmentation commands
the reference's driver writes into NVIDIA_DALI_Pipelines' module globals (Contrastive_Learning.py:601-635) into
the kernel's 32-float parameter rows, draws what DALI drew internally (RandomResizedCrop window, CoinFlip,
noise seed) from a seeded generator, and launches the ONE kernel that replaces the DALI graph
(NVIDIA_DALI_Pipelines.py:444-480)."""
import math

import numpy as np
import torch

from . import kernels as K

_RGB2YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]])
_YIQ2RGB = np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]])


def colour_matrix(hue_deg, saturation):
    """DALI ColorTwist: rotate by ``hue`` degrees and scale by ``saturation`` in the IQ plane of YIQ."""
    h = math.radians(float(hue_deg))
    rot = np.array([[1, 0, 0], [0, math.cos(h), -math.sin(h)], [0, math.sin(h), math.cos(h)]])
    sat = np.diag([1.0, float(saturation), float(saturation)])
    return _YIQ2RGB @ rot @ sat @ _RGB2YIQ


def _vec(t, b, default):
    if t is None:
        return np.full(b, default, dtype=np.float64)
    a = torch.as_tensor(t).detach().cpu().double().reshape(-1).numpy()
    if a.size == 1:
        a = np.repeat(a, b)
    assert a.size == b, "augmentation command length %d != batch %d" % (a.size, b)
    return a


def build_params(batch, src_hw, rng, pos_x=None, pos_y=None, angle=None, gm_ratio=None, gm_tile=None, noise_mean=None,
                 noise_std=None, brightness=None, contrast=None, hue=None, saturation=None, random_area=(0.1, 1.0),
                 labeled=False):
    """[B,32] float32 parameter rows.  ``src_hw``: [B,2] true (h, w) of each image inside the padded batch.
    ``labeled`` = the evaluation variant (LabeledFoveatedRetinalProcessor, :491-544): centre crop, no flip."""
    b = batch
    P = np.zeros((b, 32), dtype=np.float64)
    hw = np.asarray(src_hw, dtype=np.float64).reshape(b, 2)
    P[:, 0], P[:, 1] = hw[:, 0], hw[:, 1]
    for i in range(b):
        h, w = hw[i]
        if labeled:
            side = min(h, w)
            cw = ch = side
            x0, y0 = (w - cw) / 2, (h - ch) / 2
        else:  # RandomResizedCrop(area in random_area, aspect in [3/4, 4/3])
            area = rng.uniform(*random_area) * h * w
            ar = math.exp(rng.uniform(math.log(3 / 4), math.log(4 / 3)))
            cw, ch = min(math.sqrt(area * ar), w), min(math.sqrt(area / ar), h)
            x0, y0 = rng.uniform(0, w - cw), rng.uniform(0, h - ch)
        P[i, 2:6] = (x0, y0, cw, ch)
        P[i, 8] = 0.0 if labeled else float(rng.integers(0, 2))
        P[i, 17] = float(rng.integers(0, 1 << 23))
    ang = np.radians(_vec(angle, b, 0.0))
    P[:, 6], P[:, 7] = np.cos(ang), np.sin(ang)
    P[:, 9] = _vec(gm_ratio, b, 0.0)
    P[:, 10] = np.maximum(_vec(gm_tile, b, 1.0), 1.0)
    px, py = _vec(pos_x, b, 0.5), _vec(pos_y, b, 0.5)
    P[:, 11], P[:, 12] = px * 640.0, py * 640.0            # DALI GridMask shift_x / shift_y = the fixation
    P[:, 13], P[:, 14] = np.cos(ang), np.sin(ang)          # ... and its angle = the saccade angle (:456)
    P[:, 15], P[:, 16] = _vec(noise_mean, b, 0.0), _vec(noise_std, b, 0.0)
    br, ct = _vec(brightness, b, 1.0), _vec(contrast, b, 1.0)
    hu, sa = _vec(hue, b, 0.0), _vec(saturation, b, 1.0)
    for i in range(b):
        P[i, 18:27] = colour_matrix(hu[i], sa[i]).reshape(-1)
    P[:, 27] = br * ct
    P[:, 28] = br * 128.0 * (1.0 - ct)
    P[:, 29], P[:, 30] = np.clip(px, 0, 1), np.clip(py, 0, 1)
    return torch.from_numpy(P.astype(np.float32))


def foveate(images, params, out_size=30):
    """images [B,H,W,3] u8 on the HIP device, params [B,32] -> 4 uint8 views [B,out,out,3] (400/240/100/30 px crops)."""
    return K.foveate_views_u8(images.contiguous(), params.to(images.device).contiguous(), out_size)
