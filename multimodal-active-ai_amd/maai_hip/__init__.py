"""maai_hip — MI355X-native (gfx950) SimCLR contrastive-pretraining hot path.

Host-side Python over the C ABI in include/maai_hip.h (libmaai_hip.so, hand-written
HIP).  torch is used for device memory, streams, autograd plumbing and
torch.distributed only.
"""
from ._lib import BF16, F32, MaaiError, lib, LIB_PATH  # noqa: F401
