"""world_size-2 gloo (CPU) coverage of the N > 1 host logic: the packed embedding
all-gather, label/mask offsets by GLOBAL rank (Objective.py:55, Contrastive_Learning.py:688),
mean-of-rank-losses == global loss, and the bucketed gradient all-reduce."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "multimodal-active-ai_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import simclr_oracle as O
    from maai_hip import loss as L
    from maai_hip.dist import GradAllReduce
    torch.manual_seed(1234)
    b, d = 8, 128
    H1, H2 = torch.randn(world * b, d), torch.randn(world * b, d)
    h1, h2 = H1[rank * b:(rank + 1) * b].contiguous(), H2[rank * b:(rank + 1) * b].contiguous()
    z1, z2 = O.l2_normalize(h1), O.l2_normalize(h2)
    Z1, Z2 = L.gather_normalized(z1, z2, world)          # the product's gather on the gloo backend
    ok_gather = torch.allclose(Z1, O.l2_normalize(H1)) and torch.allclose(Z2, O.l2_normalize(H2))
    loss, g = O.nt_xent_grad_h2(h1, h2, 0.5, True, rank, world, (Z1, Z2))
    # bucketed gradient averaging
    params = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2, 2))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    GradAllReduce(params, bucket_bytes=64)()
    avg = [p.grad.flatten()[0].item() for p in params]
    q.put((rank, loss.item(), g.numpy(), ok_gather, avg))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_gloo(golden_dir, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, 29733, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    G = np.load(os.path.join(golden_dir, "ntxent_gloo.npz"))
    for r, loss, g, ok, avg in res:
        assert ok
        np.testing.assert_allclose(loss, G[f"w{world}_loss"][r], rtol=1e-6)
        np.testing.assert_allclose(g, G[f"w{world}_dh2"][r], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(avg, [1.5, 3.0, 4.5])
    np.testing.assert_allclose(np.mean([x[1] for x in res]), G[f"w{world}_global_loss"], rtol=1e-6)
