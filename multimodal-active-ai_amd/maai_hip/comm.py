"""One-shot direct all-gather over the xGMI mesh (csrc/comm.hip) for the embedding exchange of the large-negative regime
(Objective._cross_replica_concat, SimCLR/Objective.py:102-114): every rank writes its [B,d] block straight into its slot
of every peer's symmetric buffer (peer-to-peer stores, all links at once) and gathers from its own memory — one kernel,
no ring hops, no list of W tensors + cat.

The RCCL path (``torch.distributed.all_gather_into_tensor``) stays the default transport of ``maai_hip.dist``; this one
is used by ``dist.prefetch_embedding`` when MAAI_P2P_GATHER=1 AND ``P2PGather.setup`` succeeded on EVERY rank (the ranks
agree through an all-reduce, so that no rank ever issues a different collective sequence from the others)."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import MaaiError


class P2PGather(object):
    """Symmetric-buffer all-gather of messages up to ``max_bytes`` per rank.  Collective constructor: every rank of
    ``group`` must call it (handles are exchanged with ``all_gather_object``)."""

    def __init__(self, max_bytes, device=None, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise MaaiError("P2PGather needs an initialised process group")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_bytes = (int(max_bytes) + 3) // 4 * 4
        self._h = C.c_void_p()
        lib = _lib.lib()
        # Every step below is collective-safe: a rank that fails locally still takes part in the exchanges, and all
        # ranks reach the same verdict (all raise, or none does).
        payload, err = None, None
        with torch.cuda.device(self.device):
            try:
                _lib.check(lib.maai_comm_create(self.rank, self.world, self.max_bytes, C.byref(self._h)), "maai_comm_create")
                buf = C.create_string_buffer(64)
                _lib.check(lib.maai_comm_handle(self._h, buf), "maai_comm_handle")
                payload = bytes(buf.raw)
            except MaaiError as e:
                err = e
            handles = [None] * self.world
            dist.all_gather_object(handles, payload, group=group)
            if all(h is not None for h in handles):
                try:
                    for r, h in enumerate(handles):
                        if r != self.rank:
                            _lib.check(lib.maai_comm_attach(self._h, r, C.create_string_buffer(h, 64)), "maai_comm_attach")
                except MaaiError as e:
                    err = e
            elif err is None:
                err = MaaiError("P2PGather: buffer setup failed on another rank")
            oks = [None] * self.world
            dist.all_gather_object(oks, err is None, group=group)   # (also the barrier: every buffer is attached before anyone writes)
        if not all(oks):
            self.close()
            raise err if err is not None else MaaiError("P2PGather: peer attach failed on another rank")

    def gather(self, z, out=None):
        """[B, d] (any 4-byte dtype, contiguous) -> [world*B, d], on the current stream."""
        if not z.is_cuda or not z.is_contiguous() or z.element_size() != 4:
            raise MaaiError("P2PGather.gather: a contiguous 4-byte tensor on the HIP device")
        nbytes = z.numel() * 4
        if nbytes > self.max_bytes:
            raise MaaiError("P2PGather.gather: %d bytes exceed the %d the buffers were created for" % (nbytes, self.max_bytes))
        if out is None:
            out = torch.empty((self.world * z.shape[0],) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
        self.check()
        _lib.check(_lib.lib().maai_comm_allgather(self._h, C.c_void_p(z.data_ptr()), nbytes, C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "maai_comm_allgather")
        return out

    def status(self):
        """0, or the epoch of the latest gather that gave up waiting for a peer (synchronises with the device)."""
        s = C.c_uint(0)
        _lib.check(_lib.lib().maai_comm_status(self._h, C.byref(s)), "maai_comm_status")
        return int(s.value)

    def poll(self):
        """The same word without synchronising: what the gathers that have FINISHED so far reported."""
        s = C.c_uint(0)
        _lib.check(_lib.lib().maai_comm_poll(self._h, C.byref(s)), "maai_comm_poll")
        return int(s.value)

    def check(self):
        """Raise if a finished gather timed out on a peer (its output was NaN-poisoned, the ranks' epochs have diverged).
        Called before every gather and whenever ``maai_hip.dist`` hands a gathered tensor to the loss."""
        e = self.poll()
        if e:
            raise MaaiError("P2PGather: gather %d gave up waiting for a peer after MAAI_P2P_TIMEOUT_MS (default 120 s); its "
                            "output was poisoned with NaN and this communicator cannot be used again" % e)

    def close(self):
        if self._h:
            _lib.lib().maai_comm_destroy(self._h)
            self._h = C.c_void_p()


_P2P = {}   # tag -> {"obj": P2PGather or None, "tried": bool}


def p2p_gather_for(nbytes, group=None, tag="embeddings"):
    """The process-wide P2PGather of stream ``tag`` if MAAI_P2P_GATHER=1 and its setup succeeded on every rank, else None.
    The first call per tag is collective (setup + agreement); later calls return the cached decision.  One communicator PER
    STREAM of gathers ("embeddings": the prefetched gathers on the side stream; "syncbn": the statistics gathers on the compute
    stream): a communicator's epochs must be executed in the order they were issued, which only one stream guarantees."""
    if os.environ.get("MAAI_P2P_GATHER", "0") != "1":
        return None
    st = _P2P.setdefault(tag, {"obj": None, "tried": False})
    if not st["tried"]:
        st["tried"] = True
        try:
            st["obj"] = P2PGather(max(int(nbytes), 1 << 20), group=group)   # (raises on every rank, or on none)
        except MaaiError as e:
            import warnings
            warnings.warn("MAAI_P2P_GATHER=1 but the symmetric buffers (%s) could not be set up (%s): using the RCCL all-gather" % (tag, e))
    o = st["obj"]
    return o if (o is not None and nbytes <= o.max_bytes) else None
