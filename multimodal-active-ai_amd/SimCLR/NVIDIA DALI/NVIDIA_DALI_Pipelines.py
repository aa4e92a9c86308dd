"""Drop-in for the reference's ``SimCLR/NVIDIA DALI/NVIDIA_DALI_Pipelines.py`` on ROCm, where NVIDIA DALI does
not exist.  Same names and calling conventions as the contrastive / probe drivers use
(Contrastive_Learning.py:290-410,587-682):

* readers ``COCOReader`` / ``ImagenetReader`` with ``build() run() reset() reader_meta() batch_size device_id``;
* ``ImageCollector`` / ``LabelCollector`` (the driver assigns ``.data``);
* ``FixationCommand`` / ``NoiseCommand`` / ``GridMaskCommand`` / ``ColorCommand`` reading the writable module
  globals ``fixation_pos_x fixation_pos_y fixation_angle grid_mask_ratio grid_mask_tile noise_mean noise_std
  brightness contrast hue saturation`` each call, as the reference does (:108-313);
* ``UnlabeledFoveatedRetinalProcessor`` / ``LabeledFoveatedRetinalProcessor`` / ``FoveatedRetinalProcessor``;
* ``pytorch_wrapper(pipes)`` -> one list of uint8 ``[B,30,30,3]`` torch tensors per pipe;
* ``compute_shard_size(pipe, reader_name)`` (with ``int`` instead of the removed ``np.int``, :651-655).

The ~12-kernel DALI augmentation graph (:444-480) is ONE HIP kernel (maai_foveate_views_u8).  Dataset reading is
outside the hot path: files are decoded on the host with PIL (JPEG/PNG) or numpy (``.npy`` HWC uint8), or — when
``MAAI_SYNTHETIC_DATA=<count>`` is set or the directory does not exist — generated synthetically, which is what
the benchmark uses.  Exact DALI filter parity is unpinned (DALI is not installable here); the pinned contract
is oracle/simclr_oracle.py::foveate_views.
"""
import json
import os
import sys

import numpy as np
import torch

try:
    import maai_hip  # noqa: F401
except ImportError:
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, "..", ".."), os.path.join(_h, "..", "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import foveated as _fov

_IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".npy")


# ----------------------------------------------------------------------------
# readers
# ----------------------------------------------------------------------------
class _Batch(object):
    """What a reader hands to ImageCollector.data: padded uint8 images on the device + true extents."""

    def __init__(self, images, hw, labels=None):
        self.images, self.hw, self.labels = images, hw, labels


class _Reader(object):
    reader_name = "Reader"

    def __init__(self, batch_size, num_threads, device_id, shard_id, num_shards, dali_cpu=False, random_shuffle=False):
        self.batch_size, self.num_threads, self.device_id = batch_size, num_threads, device_id
        self.shard_id, self.num_shards, self.dali_cpu, self.random_shuffle = shard_id, num_shards, dali_cpu, random_shuffle
        self.seed = 15 + device_id
        self.files, self.labels, self._pos, self._built = [], [], 0, False

    # --- dataset listing (subclasses fill self.files / self.labels) ---
    def _list(self):
        raise NotImplementedError

    def build(self):
        self._list()
        n = len(self.files)
        if n == 0:
            raise RuntimeError("%s: no images found" % type(self).__name__)
        self.epoch_size = n
        self.epoch_size_padded = ((n + self.num_shards - 1) // self.num_shards) * self.num_shards
        beg = self.shard_id * self.epoch_size_padded // self.num_shards
        end = (self.shard_id + 1) * self.epoch_size_padded // self.num_shards
        self._shard = [i % n for i in range(beg, end)]          # pad_last_batch=True: wrap around
        if self.random_shuffle:
            np.random.default_rng(self.seed).shuffle(self._shard)
        self._pos, self._built = 0, True
        self._rng = np.random.default_rng(self.seed)

    def reader_meta(self):
        return {self.reader_name: dict(epoch_size=self.epoch_size, epoch_size_padded=self.epoch_size_padded,
                                       number_of_shards=self.num_shards, shard_id=self.shard_id, pad_last_batch=1,
                                       stick_to_shard=0)}

    def reset(self):
        self._pos = 0

    def _load(self, idx):
        f = self.files[idx]
        if isinstance(f, tuple):                                  # synthetic: (seed, h, w)
            seed, h, w = f
            return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        if f.endswith(".npy"):
            a = np.load(f)
        else:
            from PIL import Image
            a = np.asarray(Image.open(f).convert("RGB"))
        return np.ascontiguousarray(a, dtype=np.uint8)

    def run(self):
        if not self._built:
            raise RuntimeError("call build() first")
        idxs = [self._shard[(self._pos + k) % len(self._shard)] for k in range(self.batch_size)]
        self._pos += self.batch_size
        arrs = [self._load(i) for i in idxs]
        flips = self._rng.integers(0, 2, len(arrs))                # ops.Flip(horizontal=CoinFlip(0.5)) (:59-61)
        arrs = [a[:, ::-1] if f else a for a, f in zip(arrs, flips)]
        hw = np.array([a.shape[:2] for a in arrs])
        H, W = int(hw[:, 0].max()), int(hw[:, 1].max())
        batch = np.zeros((len(arrs), H, W, 3), dtype=np.uint8)
        for k, a in enumerate(arrs):
            batch[k, :a.shape[0], :a.shape[1]] = a
        dev = torch.device("cpu") if self.dali_cpu or not torch.cuda.is_available() else torch.device("cuda", self.device_id)
        labels = torch.tensor([self.labels[i] for i in idxs], dtype=torch.int64).reshape(-1, 1)
        return self._outputs(_Batch(torch.from_numpy(batch).to(dev), hw, labels), labels)


def _synthetic(n, seed0):
    return [(seed0 + i, 480 + 16 * (i % 5), 640 - 32 * (i % 3)) for i in range(n)], [i % 1000 for i in range(n)]


def _missing(kind, path):
    """DALI's readers fail on a dataset path that does not exist; so do these.  Random images are produced ONLY on
    request (MAAI_SYNTHETIC_DATA=<count>): a mistyped or unmounted path must never train or evaluate on noise."""
    return FileNotFoundError("%s: file_root %r is not a directory (set MAAI_SYNTHETIC_DATA=<images> to run on synthetic "
                             "images on purpose)" % (kind, path))


class COCOReader(_Reader):
    reader_name = "COCOReader"

    def __init__(self, batch_size, num_threads, device_id, file_root, annotations_file, shard_id, num_shards, dali_cpu=False):
        super().__init__(batch_size, num_threads, device_id, shard_id, num_shards, dali_cpu)
        self.file_root, self.annotations_file = file_root, annotations_file

    def _list(self):
        syn = int(os.environ.get("MAAI_SYNTHETIC_DATA", "0"))
        if syn:
            self.files, self.labels = _synthetic(syn, 1000)
            return
        if not os.path.isdir(self.file_root):
            raise _missing("COCOReader", self.file_root)
        names = None
        if os.path.isfile(self.annotations_file):
            with open(self.annotations_file) as fh:
                names = [im["file_name"] for im in json.load(fh).get("images", [])]
        if not names:
            names = sorted(f for f in os.listdir(self.file_root) if f.lower().endswith(_IMG_EXT))
        self.files = [os.path.join(self.file_root, f) for f in names]
        self.labels = [0] * len(self.files)

    def _outputs(self, batch, labels):
        return (batch, None, labels)                               # (images, bboxes, labels) (:64)


class ImagenetReader(_Reader):
    reader_name = "ImagesReader"

    def __init__(self, batch_size, num_threads, device_id, file_root, shard_id, num_shards, random_shuffle=False, dali_cpu=False):
        super().__init__(batch_size, num_threads, device_id, shard_id, num_shards, dali_cpu, random_shuffle)
        self.file_root = file_root

    def _list(self):
        syn = int(os.environ.get("MAAI_SYNTHETIC_DATA", "0"))
        if syn:
            self.files, self.labels = _synthetic(syn, 2000)
            return
        if not os.path.isdir(self.file_root):
            raise _missing("ImagenetReader", self.file_root)
        classes = sorted(d for d in os.listdir(self.file_root) if os.path.isdir(os.path.join(self.file_root, d)))
        for ci, c in enumerate(classes):                           # ops.FileReader: one label per sub-directory
            for f in sorted(os.listdir(os.path.join(self.file_root, c))):
                if f.lower().endswith(_IMG_EXT):
                    self.files.append(os.path.join(self.file_root, c, f))
                    self.labels.append(ci)

    def _outputs(self, batch, labels):
        return (batch, labels)                                     # (images, labels) (:631)


# ----------------------------------------------------------------------------
# external sources driven through module-level globals, like the reference
# ----------------------------------------------------------------------------
class ImageCollector(object):
    def __iter__(self):
        return self

    def __next__(self):
        return self.data
    next = __next__


class LabelCollector(ImageCollector):
    pass


class _Command(object):
    names, defaults = (), ()

    def __init__(self, batch_size):
        self.batch_size = batch_size

    def _get_vectors(self):
        g = globals()
        if not all(n in g for n in self.names):
            print('Initialating %s\n' % type(self).__name__)
            for n, d in zip(self.names, self.defaults):
                g[n] = d(self.batch_size)
        self.vectors = [g[n] for n in self.names]

    def __iter__(self):
        self._get_vectors()
        for v in self.vectors:
            assert len(v) == self.batch_size
        self.i, self.n = 0, len(self.vectors[0])
        return self

    def __next__(self):
        self._get_vectors()
        out = tuple([] for _ in self.vectors)
        for _ in range(self.batch_size):
            for o, v in zip(out, self.vectors):
                o.append(v[self.i])
            self.i = (self.i + 1) % self.n
        return out
    next = __next__

    def current(self):
        """the whole batch of each command vector as tensors (what the fused kernel consumes)"""
        self._get_vectors()
        return [torch.as_tensor(v).reshape(-1) for v in self.vectors]


class FixationCommand(_Command):
    names = ("fixation_pos_x", "fixation_pos_y", "fixation_angle")
    defaults = (lambda b: torch.rand((b, 1)), lambda b: torch.rand((b, 1)), lambda b: (torch.rand((b, 1)) - 0.5) * 60)


class NoiseCommand(_Command):
    names = ("noise_mean", "noise_std")
    defaults = (lambda b: torch.rand((b, 1)), lambda b: torch.rand((b, 1)))


class GridMaskCommand(_Command):
    names = ("grid_mask_ratio", "grid_mask_tile")
    defaults = (lambda b: torch.rand((b, 1)), lambda b: torch.rand((b, 1)))


class ColorCommand(_Command):
    names = ("brightness", "contrast", "hue", "saturation")
    defaults = (lambda b: torch.rand((b, 1)) * 2, lambda b: torch.rand((b, 1)) * 2, lambda b: torch.rand((b, 1)) * 360,
                lambda b: torch.rand((b, 1)))


# ----------------------------------------------------------------------------
# the foveated retinal processors
# ----------------------------------------------------------------------------
class _Processor(object):
    labeled = False

    def __init__(self, batch_size, num_threads, device_id, fixation_information, noise_information=None, color_information=None,
                 grid_mask_information=None, images=None, dali_cpu=False, labels=None):
        self.batch_size, self.num_threads, self.device_id, self.dali_cpu = batch_size, num_threads, device_id, dali_cpu
        self.fixation, self.noise, self.color, self.grid_mask = fixation_information, noise_information, color_information, grid_mask_information
        self.images, self.labels = images, labels
        self.seed = 15 + device_id
        self._rng = np.random.default_rng(self.seed)
        self._outs = None

    def build(self):
        self._rng = np.random.default_rng(self.seed)

    def _batch(self):
        data = next(iter(self.images)) if not hasattr(self.images, "data") else self.images.data
        if isinstance(data, _Batch):
            return data
        t = torch.as_tensor(data)
        return _Batch(t, np.array([t.shape[1:3]] * t.shape[0]))

    def run(self):
        batch = self._batch()
        b = batch.images.shape[0]
        px, py, ang = self.fixation.current()
        kw = dict(pos_x=px, pos_y=py, angle=ang, labeled=self.labeled)
        if self.noise is not None:
            kw["noise_mean"], kw["noise_std"] = self.noise.current()
        if self.grid_mask is not None:
            kw["gm_ratio"], kw["gm_tile"] = self.grid_mask.current()
        if self.color is not None:
            kw["brightness"], kw["contrast"], kw["hue"], kw["saturation"] = self.color.current()
        params = _fov.build_params(b, batch.hw, self._rng, **kw)
        views = _fov.foveate(batch.images, params)                 # ONE kernel: 4 x [B,30,30,3] uint8
        if self.labeled:
            views = views + [batch.labels.to(views[0].device) if batch.labels is not None else None]
        self._outs = views
        return views

    # DALI pipeline protocol used by pytorch_wrapper (:553-581)
    def schedule_run(self):
        self.run()

    def share_outputs(self):
        return self._outs

    def release_outputs(self):
        self._outs = None

    def reset(self):
        pass


class UnlabeledFoveatedRetinalProcessor(_Processor):
    """training augmentation (:400-480)"""


class FoveatedRetinalProcessor(_Processor):
    """plotting variant (:316-388): no grid mask"""

    def __init__(self, batch_size, num_threads, device_id, fixation_information, noise_information, color_information, images,
                 dali_cpu=False):
        super().__init__(batch_size, num_threads, device_id, fixation_information, noise_information, color_information, None, images, dali_cpu)


class LabeledFoveatedRetinalProcessor(_Processor):
    """evaluation variant (:491-544): centre crop, no random flip; also returns the labels"""
    labeled = True

    def __init__(self, batch_size, num_threads, device_id, fixation_information, images, labels=None, dali_cpu=False):
        super().__init__(batch_size, num_threads, device_id, fixation_information, None, None, None, images, dali_cpu, labels)


def pytorch_wrapper(pipes):
    """One list of torch tensors per pipe (the reference copies DALI buffers into torch tensors here; the HIP kernel
    already wrote torch tensors on torch's current stream, so this is just the hand-over)."""
    outs = []
    for p in pipes:
        p.schedule_run()
    for p in pipes:
        outs.append(list(p.share_outputs()))
        p.release_outputs()
    return outs


def compute_shard_size(pipe, reader_name):
    meta = pipe.reader_meta()[reader_name]
    size = meta['epoch_size_padded'] if meta['pad_last_batch'] == 1 else meta['epoch_size']
    beg = int(np.floor(meta['shard_id'] * size / meta['number_of_shards']))
    end = int(np.floor((meta['shard_id'] + 1) * size / meta['number_of_shards']))
    return end - beg
