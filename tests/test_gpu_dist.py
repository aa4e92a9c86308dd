"""Two (and five) ranks on the N > 1 code path: SyncBatchNorm statistics (the projection-shortcut pair in one exchange), the
embedding all-gather prefetched on the side stream by SimCLR_Module.forward, the rank-offset labels and the gradient
buckets all-reduced from inside the backward pass must reproduce the single-process result on the concatenated
batch (the reference's multi-rank semantics, Objective.py:51-58).
* test_two_ranks_match_single_process: both ranks share ONE MI355X over gloo (runs on the one-GPU box);
* test_two_ranks_rccl: one GPU per rank over the 'nccl' (= RCCL) backend; skipped where fewer than two GPUs exist."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    pkg = os.path.join(ROOT, "multimodal-active-ai_amd")
    sim = os.path.join(pkg, "SimCLR")
    for d in (ROOT, pkg, sim, os.path.join(sim, "ResNet"), os.path.join(sim, "MLP")):
        if d not in sys.path:
            sys.path.insert(0, d)


def _model(world, B, arch="resnet18"):
    _paths()
    import resnet as rn
    import multilayerPerceptron as mlp
    import SimCLR
    from oracle import simclr_oracle as O
    norm = torch.nn.SyncBatchNorm if world > 1 else torch.nn.BatchNorm2d
    head_in = 512 * O.expansion(arch) * 16
    m = SimCLR.SimCLR_Module(getattr(rn, arch)(crop_measures=1, norm_layer=norm), mlp.MLP(head_in, 1024, 128), B, (32, 32), "cuda")
    m.load_state_dict(O.pattern_state_dict(arch, 1, head_in, residual_gamma=0.25), strict=True)
    return m.cuda().train()


def _inputs(B):
    g = torch.Generator().manual_seed(5)
    return (torch.randint(0, 256, (B, 3, 32, 32), generator=g).float(), torch.randint(0, 256, (B, 3, 32, 32), generator=g).float())


def _worker(rank, world, port, q, backend="gloo", arch="resnet18", p2p=False, auto=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MAAI_P2P_GATHER"] = "1" if p2p else "0"
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    import Objective
    from maai_hip import engine
    from maai_hip import dist as D
    from maai_hip.dist import GradReducer
    engine.set_precision("fp32")
    engine.set_grad_allreduce("0")             # step A wants this rank's LOCAL gradient (the reference's behaviour)
    B = 16 if world == 2 else 4 * world
    b = B // world
    x1, x2 = _inputs(B)
    m = _model(world, b, arch)
    if arch == "resnet50":
        engine.set_lazy(True, True, "all")     # every lazy / join launch under SyncBatchNorm, not only the policy's
    sl = slice(rank * b, (rank + 1) * b)
    # step A: no gradient exchange -> the local gradient of this rank
    with torch.no_grad():
        h1 = m.forward_tensor(x1[sl].cuda())
    h2 = m.forward_tensor(x2[sl].cuda())
    loss, logits, labels = Objective.contrastive_loss(h1.data, h2, temperature=0.5, local_rank=rank, world_size=world, device="cuda")
    loss.backward()
    torch.cuda.synchronize()
    g_local = {n: p.grad.clone() for n, p in m.named_parameters()
               if n in ("f.conv1.weight", "g.layers.2.bias", "f.layer3.0.bn2.weight", "f.layer2.0.downsample.1.weight")}
    rm = m.f.bn1.running_mean.cpu().numpy()
    hits_a = D.STATS["prefetch_hits"]
    # step B, same weights and inputs: gradient buckets go out from inside the backward pass (engine hook)
    m.zero_grad(set_to_none=True)
    if auto:
        # the UNCHANGED driver never installs a hook (Contrastive_Learning.py:418-424 unwraps DDP): the engine finds the
        # process group at the first backward and builds the reducer itself (MAAI_GRAD_ALLREDUCE=auto, the default)
        engine.set_grad_allreduce("auto")
        red = None
    else:
        red = GradReducer(list(m.parameters()), bucket_bytes=16 << 20)
        engine.set_grad_hook(red)
    with torch.no_grad():
        h1b = m.forward_tensor(x1[sl].cuda())
    h2b = m.forward_tensor(x2[sl].cuda())
    lossb, _, _ = Objective.contrastive_loss(h1b.data, h2b, temperature=0.5, local_rank=rank, world_size=world, device="cuda")
    lossb.backward()
    torch.cuda.synchronize()
    engine.set_grad_hook(None)
    if auto:
        assert len(engine._AUTO_REDUCE["hooks"]) == 1
        red = next(iter(engine._AUTO_REDUCE["hooks"].values()))[0]
    g_avg = {n: p.grad.clone() for n, p in m.named_parameters() if n in g_local}
    stats = dict(D.STATS, launched_early=red.launched_early, nbuckets=len(red.buckets), hits_a=hits_a, exchanges=dict(engine.EXCHANGES))
    q.put((rank, loss.item(), h2.detach().cpu().numpy(), logits.cpu().numpy(), labels.argmax(1).cpu().numpy(), rm,
           {n: v.cpu().numpy() for n, v in g_local.items()}, {n: v.cpu().numpy() for n, v in g_avg.items()}, stats))
    dist.destroy_process_group()


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_two_ranks_rccl(arch):
    """The same protocol with one MI355X per rank over RCCL (backend 'nccl')."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: this box has %d" % torch.cuda.device_count())
    _run_two_ranks("nccl", 29743, arch)


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_two_ranks_match_single_process(arch):
    """resnet50: Bottleneck blocks — the projection-shortcut pair's statistics in one exchange, forward and backward,
    and every normalise-on-load / join launch fed by all-reduced statistics."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    _run_two_ranks("gloo", 29741 if arch == "resnet18" else 29745, arch)


def test_two_ranks_direct_allgather_transport():
    """The same step with the embedding gathers on the one-shot direct all-gather (csrc/comm.hip: symmetric buffers opened
    across processes by IPC handle, peer-to-peer {epoch, payload} granule stores, MAAI_P2P_GATHER=1): same embeddings, loss,
    labels and gradients; all four gathers of the two steps took the direct transport."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    _run_two_ranks("gloo", 29751, "resnet18", p2p=True)


def test_two_ranks_gradient_allreduce_installs_itself():
    """under the unchanged driver nobody calls engine.set_grad_hook — the reference unwraps DDP
    (Contrastive_Learning.py:418-424) and would let the replicas drift.  With the default MAAI_GRAD_ALLREDUCE=auto the first
    backward that finds a process group of two ranks builds the bucketed reducer itself: both ranks end with the average of
    the two local gradients, buckets leave from inside the backward pass."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    _run_two_ranks("gloo", 29757, "resnet18", auto=True)


def test_five_ranks_on_one_device():
    """The N > 1 protocol beyond two ranks, as far as ONE device allows: the GPU box admits six processes on the card, one of
    them this test process, hence five ranks.  SyncBatchNorm gathers of five rows (Chan merge), five-slot direct all-gather
    (csrc/comm.hip, both epoch parities over the two steps), the reducer's presence words at five ranks, exchange counts."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    _run_two_ranks("gloo", 29759, "resnet18", p2p=True, world=5)


def _p2p_timeout_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MAAI_P2P_TIMEOUT_MS"] = "400"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    from maai_hip.comm import P2PGather
    from maai_hip._lib import MaaiError
    torch.cuda.set_device(0)
    try:
        pg = P2PGather(1 << 16)
        z = torch.full((8, 16), float(rank + 1), device="cuda")
        out = pg.gather(z)                      # epoch 1: both ranks take part
        torch.cuda.synchronize()
        ok1 = bool(torch.equal(out, torch.cat([torch.full((8, 16), float(r + 1)) for r in range(world)]).cuda())) and pg.poll() == 0
        dist.barrier()
        res = None
        if rank == 0:
            out2 = pg.gather(z)                 # epoch 2: rank 1 never writes — the sweep must give up after 0.4 s ...
            torch.cuda.synchronize()
            mine_ok = bool(torch.equal(out2[:8], z))
            theirs_nan = bool(torch.isnan(out2[8:]).all())       # ... deliver NaN, never stale epoch-1 data ...
            polled = pg.poll()                                   # ... say so in the host-visible status word ...
            raised = False
            try:
                pg.gather(z)                                     # ... and refuse to be used again
            except MaaiError:
                raised = True
            res = (ok1, mine_ok, theirs_nan, polled, raised)
        dist.barrier()
        q.put((rank, res if rank == 0 else (ok1,)))
        pg.close()
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    dist.destroy_process_group()


def test_direct_allgather_timeout_is_loud():
    """a sweep that gives up on a peer used to copy whatever the buffer held into the output and set a word nobody
    read.  Now: time-based (MAAI_P2P_TIMEOUT_MS) give-up, NaN instead of stale data, a host-visible status word that
    P2PGather.check() / the next gather turn into an exception."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_p2p_timeout_worker, args=(r, 2, 29761, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=240) for _ in ps)
    [p.join(60) for p in ps]
    assert res[1] == (True,), res
    ok1, mine_ok, theirs_nan, polled, raised = res[0]
    assert ok1 and mine_ok and theirs_nan and polled == 2 and raised, res


def _p2p_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    from maai_hip.comm import P2PGather
    torch.cuda.set_device(0)
    ok, detail = True, []
    try:
        pg = P2PGather(1 << 20)
        for trial, b in enumerate([16, 512, 2048, 33, 512, 1, 2048, 512]):   # 8 KB .. 1 MB per rank, both epoch parities
            g = torch.Generator().manual_seed(100 * trial + rank)
            z = torch.randn(b, 128, generator=g).cuda()
            out = pg.gather(z)
            ref = torch.empty(world * b, 128, device="cuda")
            dist.all_gather_into_tensor(ref, z)          # (also the flow control a real step provides between gathers)
            torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            ok = ok and same
            detail.append((b, same))
        st = pg.status()
        ok = ok and st == 0
        # an integer payload goes through bit for bit too (the labels / counts some callers send)
        zi = (torch.arange(64 * 4, device="cuda", dtype=torch.int32) * (rank + 3)).reshape(64, 4)
        oi = pg.gather(zi)
        torch.cuda.synchronize()
        exp = torch.cat([(torch.arange(64 * 4, dtype=torch.int32) * (r + 3)).reshape(64, 4) for r in range(world)]).cuda()
        ok = ok and bool(torch.equal(oi, exp))
        pg.close()
        q.put((rank, ok, detail, st))
    except Exception as e:   # noqa: BLE001
        q.put((rank, False, repr(e), -1))
    dist.destroy_process_group()


def test_direct_allgather_matches_process_group_gather():
    """csrc/comm.hip on its own: two processes on one MI355X, buffers exchanged by hipIpc handles, eight gathers of 0.5 KB to
    1 MB per rank (alternating epoch parities) against torch.distributed's all_gather; no sweep gave up on a peer."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_p2p_worker, args=(r, 2, 29753, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    for r, ok, detail, st in res:
        assert ok and st == 0, (r, detail, st)


def _run_two_ranks(backend, port, arch="resnet18", p2p=False, world=2, auto=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, backend, arch, p2p, auto)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    # single process on the concatenated batch
    _paths()
    import Objective
    from maai_hip import engine
    engine.set_precision("fp32")
    B = 16 if world == 2 else 4 * world
    x1, x2 = _inputs(B)
    m = _model(1, B, arch)
    with torch.no_grad():
        h1 = m.forward_tensor(x1.cuda())
        h2 = m.forward_tensor(x2.cuda())
        loss, logits, _ = Objective.contrastive_loss(h1, h2, temperature=0.5)
    engine.set_precision("bf16")
    b = B // world
    for r, l, z, lg, lab, rm, gl, ga, st in res:
        # SyncBN statistics are global -> per-rank embeddings equal the big-batch ones
        np.testing.assert_allclose(z, h2[r * b:(r + 1) * b].cpu().numpy(), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(lg, logits[r * b:(r + 1) * b].cpu().numpy(), rtol=2e-3, atol=2e-3)
        assert (lab == np.arange(b) + r * b).all()
        np.testing.assert_allclose(rm, m.f.bn1.running_mean.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # mean over ranks of the per-rank loss == single-process loss on the global batch
    np.testing.assert_allclose(np.mean([x[1] for x in res]), loss.item(), rtol=1e-4)
    # gradient exchange from inside the backward pass: both ranks hold the average of the two local gradients
    for name in res[0][6]:
        avg = sum(x[6][name] for x in res) / world
        for x in res:
            np.testing.assert_allclose(x[7][name], avg, rtol=2e-5, atol=1e-6 * float(np.abs(avg).max()))
    for x in res:
        st = x[8]
        # every embedding gather was started by the forward pass on the side stream and picked up by the loss
        assert st["prefetch_started"] == 4 and st["prefetch_hits"] == 4 and st["hits_a"] == 2, st
        # the head's bucket(s) went out before the backbone was differentiated
        assert st["nbuckets"] >= (2 if auto else 3) and st["launched_early"] >= 1 and st["buckets_early"] + st["buckets_late"] == st["nbuckets"], st
        # SyncBatchNorm exchanges (fp32 mean | M2 | count gathers forward, fp32 sum all-reduces backward): one per
        # BatchNorm layer, the two that meet at a projection shortcut sharing one — two steps of (no-grad forward,
        # forward, backward) were run: the count DESIGN section 6 budgets latency for
        per_pass = 17 if arch == "resnet18" else 49
        ex = dict(st["exchanges"])
        # (MAAI_P2P_GATHER=1: the SyncBatchNorm forward gathers ride the direct all-gather too — every one of them)
        assert ex.pop("p2p", 0) == (4 * per_pass if p2p else 0), st["exchanges"]
        assert ex == {"gather": 4 * per_pass, "reduce": 2 * per_pass}, st["exchanges"]
        assert st.get("p2p_gathers", 0) == (4 if p2p else 0), st
