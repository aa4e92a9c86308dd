// One-shot DIRECT all-gather of small per-rank messages over the xGMI mesh: the embedding exchange of
// the large-negative regime — Objective._cross_replica_concat (SimCLR/Objective.py:102-114: dist.all_gather into a list of
// W tensors + torch.cat), [B,128] fp32 = 256 KB per rank at B = 512 — and messages of that class.
//
// xGMI is point to point: every GPU has a dedicated link to each of its 7 peers, so a small all-gather needs no ring and
// no hops — every rank WRITES its message straight into its slot of every peer's buffer, all links at once, and then
// reads only its own memory.  Each rank owns one symmetric buffer (hipExtMallocWithFlags, fine-grained; exported with
// hipIpcGetMemHandle and opened by the peers) of 2 (epoch parity) x world x slot granules.  A granule is ONE naturally
// aligned 8-byte {epoch tag, 32-bit payload} written by ONE system-scope store: the data is the flag — no separate
// flag, no fence, nothing to order (cdna_hip_programming.md Guideline 16, form R2, here across devices).  The gather is
// one kernel: (1) every thread turns its share of the message into granules and stores them to all `world` buffers
// (its own included); (2) every thread sweeps its share of the LOCAL buffer until each granule carries this call's
// epoch and copies the payloads to the output.  Spins are bounded IN TIME (wall clock, MAAI_P2P_TIMEOUT_MS, default 120 s:
// a rank-0 checkpoint or a first-step autotune may legitimately skew the ranks by seconds): a peer that never writes makes
// the kernel give up instead of hanging the GPU — and a give-up is never silent: the granule is delivered as a quiet NaN
// (the loss of that step is NaN on the rank that timed out) and a status word in host-visible memory is set, which
// maai_comm_allgather checks before every launch (MAAI_ERR_LAUNCH from then on: the epochs of the ranks no longer agree)
// and maai_comm_poll / maai_comm_status return.
// Flow control is the caller's: a rank may run at most ONE gather ahead of any other rank (the two parities); the SimCLR
// step guarantees it (the gradient all-reduce and the SyncBatchNorm exchanges of a step sit between its two gathers and
// the next step's).  The RCCL path (torch.distributed all_gather_into_tensor) stays the default transport of
// maai_hip.dist; this one is selected with MAAI_P2P_GATHER=1 after every rank has confirmed its buffers are attached.
#include "common.h"
#include "maai_internal.h"
#include <stdlib.h>
#include <string.h>

struct maai_comm {
  int rank, world;
  long long slot_granules;        // capacity of one rank's slot, in granules (= message bytes / 4)
  unsigned long long* local;      // this rank's buffer: [2][world][slot_granules]
  unsigned long long* peer[64];   // peer[r]: rank r's buffer as mapped here (peer[rank] = local)
  unsigned long long** peer_dev;  // device copy of peer[]
  unsigned* status_host;          // pinned, mapped host word: 0 ok, else the epoch at which a sweep gave up
  unsigned* status_dev;           // the same word as the device sees it
  unsigned long long timeout_ticks;   // wall-clock ticks (100 MHz) a sweep waits for one granule
  unsigned epoch;
  int attached;
};

typedef __attribute__((address_space(1))) unsigned long long gu64_t;

__global__ __launch_bounds__(256) void comm_allgather_kernel(unsigned long long* const* __restrict__ peers, const unsigned long long* local,
                                                             const unsigned* __restrict__ src, unsigned* __restrict__ dst, long long granules,
                                                             long long slot_granules, int rank, int world, unsigned epoch, unsigned* status,
                                                             unsigned long long timeout_ticks) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nth = (long long)gridDim.x * blockDim.x;
  const long long par = (long long)(epoch & 1u) * world * slot_granules;
  // (1) publish: one 8-byte system-scope store per granule and peer
  for (long long g = tid; g < granules; g += nth) {
    const unsigned long long v = ((unsigned long long)epoch << 32) | (unsigned long long)src[g];
    for (int p = 0; p < world; ++p) {
      unsigned long long* q = peers[p] + par + (long long)rank * slot_granules + g;
      __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // (2) sweep the local buffer: every rank's slot, until each granule carries this epoch
  const long long total = (long long)world * granules;
  for (long long i = tid; i < total; i += nth) {
    const int r = (int)(i / granules);
    const long long g = i - (long long)r * granules;
    const unsigned long long* q = local + par + (long long)r * slot_granules + g;
    unsigned long long v = 0;
    unsigned long long t0 = 0;
    unsigned spins = 0;
    for (;;) {
      v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned)(v >> 32) == epoch) break;
      if ((++spins & 1023u) == 0) {   // the constant-rate wall clock, read once per ~thousand polls
        const unsigned long long now = wall_clock64();
        if (t0 == 0) {
          t0 = now;
        } else if (now - t0 > timeout_ticks) {
          // a peer is gone (or hopelessly late).  Give up LOUDLY instead of hanging the device: the status word (host
          // visible, system scope) names the epoch and the granule is delivered as a quiet NaN, never as stale data.
          __hip_atomic_store(status, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          v = 0x7fc00000ull;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(8);
    }
    dst[i] = (unsigned)v;
  }
}

extern "C" int maai_comm_create(int rank, int world, long long max_bytes, maai_comm** out) {
  MAAI_CHECK_ARG(out && world >= 1 && world <= 64 && rank >= 0 && rank < world && max_bytes > 0 && max_bytes % 4 == 0,
                 "comm_create: bad arguments");
  maai_comm* c = (maai_comm*)calloc(1, sizeof(maai_comm));
  if (!c) {
    maai_set_error("comm_create: out of host memory");
    return MAAI_ERR_LAUNCH;
  }
  c->rank = rank;
  c->world = world;
  c->slot_granules = max_bytes / 4;
  const size_t bytes = (size_t)2 * world * c->slot_granules * 8;
  void* p = nullptr;
  // fine-grained memory only: the sweep polls words that OTHER devices write, which coarse-grained memory does not keep
  // coherent — no fallback (the caller then stays on the RCCL all-gather)
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    maai_set_error("comm_create: fine-grained device memory (hipDeviceMallocFinegrained) is not available");
    free(c);
    return MAAI_ERR_LAUNCH;
  }
  void* sh = nullptr;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipMalloc((void**)&c->peer_dev, 64 * sizeof(void*)) != hipSuccess ||
      hipHostMalloc(&sh, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->status_dev, sh, 0) != hipSuccess) {
    (void)hipGetLastError();
    maai_set_error("comm_create: device allocation failed");
    (void)hipFree(p);
    free(c);
    return MAAI_ERR_LAUNCH;
  }
  c->status_host = (unsigned*)sh;
  memset(sh, 0, 64);
  {
    const char* t = getenv("MAAI_P2P_TIMEOUT_MS");
    double ms = t ? atof(t) : 120000.0;
    if (!(ms >= 1.0)) ms = 1.0;
    c->timeout_ticks = (unsigned long long)(ms * 1e5);   // wall_clock64(): 100 MHz
  }
  c->local = (unsigned long long*)p;
  c->peer[rank] = c->local;
  c->attached = 1;
  c->epoch = 0;
  *out = c;
  return MAAI_OK;
}

// 64-byte hipIpcMemHandle_t of this rank's buffer, to be sent to every peer (any transport: the caller's process group)
extern "C" int maai_comm_handle(maai_comm* c, void* handle64) {
  MAAI_CHECK_ARG(c && handle64, "comm_handle: null pointer");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, c->local) != hipSuccess) {
    maai_set_error("comm_handle: hipIpcGetMemHandle failed (the driver exports device memory by dmabuf: HSA_ENABLE_IPC_MODE_LEGACY=0)");
    (void)hipGetLastError();
    return MAAI_ERR_LAUNCH;
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
  memcpy(handle64, &h, 64);
  return MAAI_OK;
}

extern "C" int maai_comm_attach(maai_comm* c, int peer, const void* handle64) {
  MAAI_CHECK_ARG(c && handle64 && peer >= 0 && peer < c->world, "comm_attach: bad arguments");
  if (peer == c->rank) return MAAI_OK;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    maai_set_error("comm_attach: hipIpcOpenMemHandle failed");
    (void)hipGetLastError();
    return MAAI_ERR_LAUNCH;
  }
  c->peer[peer] = (unsigned long long*)p;
  c->attached += 1;
  return MAAI_OK;
}

// dst[world][bytes] <- every rank's src[bytes] (bytes % 4 == 0, <= the max_bytes of maai_comm_create), on `stream`.
extern "C" int maai_comm_allgather(maai_comm* c, const void* src, long long bytes, void* dst, void* stream) {
  MAAI_CHECK_ARG(c && src && dst && bytes > 0 && bytes % 4 == 0 && bytes / 4 <= c->slot_granules, "comm_allgather: bad arguments");
  MAAI_CHECK_ARG(c->attached == c->world, "comm_allgather: not every peer buffer is attached");
  if (*(volatile unsigned*)c->status_host != 0) {   // an earlier gather gave up: its output was NaN and the ranks' epochs have diverged
    maai_set_error("comm_allgather: an earlier gather timed out waiting for a peer (MAAI_P2P_TIMEOUT_MS); this communicator is dead");
    return MAAI_ERR_LAUNCH;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (c->epoch == 0) {   // first call: the peer table goes to the device once
    if (hipMemcpyAsync(c->peer_dev, c->peer, c->world * sizeof(void*), hipMemcpyHostToDevice, st) != hipSuccess) {
      maai_set_error("comm_allgather: peer table upload failed");
      return MAAI_ERR_LAUNCH;
    }
  }
  c->epoch += 1;
  if (c->epoch == 0) c->epoch = 1;   // (tag 0 = never written)
  const long long granules = bytes / 4;
  long long blocks = ((long long)c->world * granules + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(comm_allgather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c->peer_dev, c->local, (const unsigned*)src,
                     (unsigned*)dst, granules, c->slot_granules, c->rank, c->world, c->epoch, c->status_dev, c->timeout_ticks);
  MAAI_CHECK_LAUNCH();
  return MAAI_OK;
}

// 0, or the epoch of the latest gather whose sweep gave up — after waiting for everything queued on the device
extern "C" int maai_comm_status(maai_comm* c, unsigned* status) {
  MAAI_CHECK_ARG(c && status, "comm_status: null pointer");
  if (hipDeviceSynchronize() != hipSuccess) {
    maai_set_error("comm_status: device synchronisation failed");
    return MAAI_ERR_LAUNCH;
  }
  *status = *(volatile unsigned*)c->status_host;
  return MAAI_OK;
}

// the same word WITHOUT synchronising (host-visible memory): what finished gathers have reported so far — cheap enough for
// the hot path (maai_hip.dist checks it whenever it hands out a gathered tensor)
extern "C" int maai_comm_poll(maai_comm* c, unsigned* status) {
  MAAI_CHECK_ARG(c && status, "comm_poll: null pointer");
  *status = *(volatile unsigned*)c->status_host;
  return MAAI_OK;
}

extern "C" int maai_comm_destroy(maai_comm* c) {
  if (!c) return MAAI_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  (void)hipFree(c->local);
  (void)hipFree(c->peer_dev);
  (void)hipHostFree(c->status_host);
  free(c);
  return MAAI_OK;
}
