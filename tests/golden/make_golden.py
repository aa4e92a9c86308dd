"""Generate golden vectors from the REFERENCE's own modules (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference)
Writes tests/golden/*.npz — inputs are regenerated from seeds/recipes and the
weights from oracle.simclr_oracle.pattern_state_dict (closed form), so only
outputs are stored.  The reference's source never enters the repo.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MAAI_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import simclr_oracle as O  # noqa: E402

# apex is not installed: Model_Util.py:6 imports it at module top
apex = types.ModuleType("apex"); par = types.ModuleType("apex.parallel"); larc = types.ModuleType("apex.parallel.LARC")
larc.LARC = object; par.LARC = larc; apex.parallel = par
sys.modules.update({"apex": apex, "apex.parallel": par, "apex.parallel.LARC": larc})
for d in ("SimCLR", "SimCLR/ResNet", "SimCLR/MLP"):
    sys.path.append(os.path.join(REF, d))
import resnet as rn  # noqa: E402
import multilayerPerceptron as mlp  # noqa: E402
import SimCLR  # noqa: E402
import Objective  # noqa: E402
import Model_Util  # noqa: E402

torch.set_num_threads(8)


def inputs_uniform_u8(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


def build_ref(arch, cm, head_in, batch, img, rg=1.0):
    f = getattr(rn, arch)(crop_measures=cm)
    g = mlp.MLP(head_in, 1024, 128)
    m = SimCLR.SimCLR_Module(f, g, batch, img, "cpu")
    m.load_state_dict(O.pattern_state_dict(arch, cm, head_in, residual_gamma=rg), strict=True)
    return m


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in kw.items()})
    print("wrote", name, {k: np.asarray(v.detach() if torch.is_tensor(v) else v).shape for k, v in kw.items()})


# ---- 2. NT-Xent 2-rank gloo ---------------------------------------------------
def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)
    b, d = 8, 128
    H1, H2 = torch.randn(world * b, d), torch.randn(world * b, d)
    h1 = H1[rank * b:(rank + 1) * b].clone()
    h2 = H2[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    loss, logits, labels = Objective.contrastive_loss(h1, h2, temperature=0.5, local_rank=rank, world_size=world, device="cpu")
    loss.backward()
    q.put((rank, loss.item(), logits.detach().numpy(), h2.grad.numpy(), labels.argmax(1).numpy()))
    dist.destroy_process_group()


def gloo_case(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, 29611 + world, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get() for _ in ps], key=lambda t: t[0])
    [p.join() for p in ps]
    return res


if __name__ == "__main__":
    # ---- 1. NT-Xent single process ------------------------------------------------
    out = {}
    for tag, (b, d, tau, seed) in {"b8": (8, 128, 0.5, 0), "b16": (16, 128, 0.5, 0), "b64": (64, 128, 0.05, 3), "b33": (33, 96, 0.2, 5)}.items():
        torch.manual_seed(seed)
        h1, h2 = torch.randn(b, d), torch.randn(b, d)
        h2r = h2.clone().requires_grad_(True)
        loss, logits, labels = Objective.contrastive_loss(h1, h2r, temperature=tau)
        loss.backward()
        out.update({f"{tag}_cfg": np.array([b, d, tau, seed]), f"{tag}_loss": loss, f"{tag}_logits": logits,
                    f"{tag}_labels_argmax": labels.argmax(1), f"{tag}_dh2": h2r.grad})
        # both sides requiring grad (validate() / generic autograd use)
        a = h1.clone().requires_grad_(True); bb = h2.clone().requires_grad_(True)
        l2, _, _ = Objective.contrastive_loss(a, bb, temperature=tau); l2.backward()
        out.update({f"{tag}_dh1_both": a.grad, f"{tag}_dh2_both": bb.grad})
        # hidden_norm False
        l3, lg3, _ = Objective.contrastive_loss(h1 * 0.1, h2 * 0.1, hidden_norm=False, temperature=tau)
        out.update({f"{tag}_loss_nonorm": l3})
    save("ntxent_single.npz", **out)


    out = {}
    for world in (2, 4):
        res = gloo_case(world)
        out[f"w{world}_loss"] = np.array([r[1] for r in res])
        out[f"w{world}_logits"] = np.stack([r[2] for r in res])
        out[f"w{world}_dh2"] = np.stack([r[3] for r in res])
        out[f"w{world}_labels_argmax"] = np.stack([r[4] for r in res])
        torch.manual_seed(1234)
        H1, H2 = torch.randn(world * 8, 128), torch.randn(world * 8, 128)
        lg, _, _ = Objective.contrastive_loss(H1, H2, temperature=0.5)
        out[f"w{world}_global_loss"] = lg
    save("ntxent_gloo.npz", **out)

    # ---- 3. ResNet-18, cfg1-shaped (3x32x32), B=16: forward, loss, grads, BN buffers, 3-step Adam ----
    B = 16
    x1 = inputs_uniform_u8(100, (B, 3, 32, 32)).float()
    x2 = inputs_uniform_u8(101, (B, 3, 32, 32)).float()
    m = build_ref("resnet18", 1, 512 * 16, B, (32, 32), 0.25); m.train()
    opt = torch.optim.Adam(m.parameters(), 1e-3)
    traj, first = [], {}
    with torch.no_grad():
        h1 = m.g(m.f(x1))
    for step in range(3):
        h2 = m.g(m.f(x2))
        loss, logits, labels = Objective.contrastive_loss(hidden1=h1.data, hidden2=h2, temperature=0.5)
        opt.zero_grad(); loss.backward()
        if step == 0:
            sd = m.state_dict()
            first = dict(z1=h1, z2=h2, loss=loss, logits=logits,
                         g_conv1=m.f.conv1.weight.grad, g_bn1_w=m.f.bn1.weight.grad, g_bn1_b=m.f.bn1.bias.grad,
                         g_l4_conv2=m.f.layer4[1].conv2.weight.grad[:16, :16].clone(), g_l2_ds=m.f.layer2[0].downsample[0].weight.grad,
                         g_fc2_w=m.g.layers[2].weight.grad, g_fc2_b=m.g.layers[2].bias.grad,
                         g_fc1_b=m.g.layers[0].bias.grad,
                         gnorms=np.array([p.grad.norm().item() for p in m.parameters()]),
                         bn1_rm=sd["f.bn1.running_mean"].clone(), bn1_rv=sd["f.bn1.running_var"].clone(),
                         l4_bn2_rm=sd["f.layer4.1.bn2.running_mean"].clone(), l4_bn2_rv=sd["f.layer4.1.bn2.running_var"].clone(),
                         nbt=sd["f.bn1.num_batches_tracked"].clone())
        opt.step()
        traj.append(loss.item())
        h1 = h2
    save("r18_cfg1.npz", traj=np.array(traj), conv1_after=m.f.conv1.weight.detach()[:4], **first)

    # ---- 4. ResNet-50 native geometry via SimCLR_Module.forward: 4 u8 HWC views, 12x30x30, B=8 ----
    B = 8
    views = [inputs_uniform_u8(200 + k, (B, 30, 30, 3)) for k in range(4)]
    m = build_ref("resnet50", 4, 2048 * 16, B, (30, 30), 0.25); m.train()
    z = m(views)
    sd = m.state_dict()
    m.eval()
    with torch.no_grad():
        z_eval = m(views)
    save("r50_native.npz", z=z, z_eval=z_eval, l1_bn3_rm=sd["f.layer1.0.bn3.running_mean"], l1_bn3_rv=sd["f.layer1.0.bn3.running_var"])

    # ---- 4b. ResNet-50 at 3x64x64 with the 4x4 adaptive pool head (the cfg2 topology, small) ----
    B = 4
    x = inputs_uniform_u8(300, (B, 3, 64, 64)).float()
    m = build_ref("resnet50", 1, 2048 * 16, B, (64, 64), 0.25); m.train()
    feat = m.f(x)
    z = m.g(torch.nn.functional.adaptive_avg_pool2d(feat, (4, 4)))
    save("r50_pool.npz", z=z, feat_mean=feat.mean(dim=(2, 3)))

    # ---- 4c. ResNet-50 AT THE TIMED RESOLUTION (BASELINE configs[1] topology): 3x224x224, reference stem, 4x4 adaptive pool
    #          (resnet.py:181's commented variant), MLP(32768,1024,128); 2 uint8 images, train-mode BatchNorm ----
    B = 2
    x = inputs_uniform_u8(400, (B, 3, 224, 224)).float()
    m = build_ref("resnet50", 1, 2048 * 16, B, (224, 224), 0.25); m.train()
    feat = m.f(x)
    z = m.g(torch.nn.functional.adaptive_avg_pool2d(feat, (4, 4)))
    sd = m.state_dict()
    save("r50_224.npz", z=z, feat_mean=feat.mean(dim=(2, 3)), feat_shape=np.array(feat.shape),
         l1_bn3_rm=sd["f.layer1.0.bn3.running_mean"], l1_bn3_rv=sd["f.layer1.0.bn3.running_var"],
         l4_bn3_rm=sd["f.layer4.2.bn3.running_mean"], l4_bn3_rv=sd["f.layer4.2.bn3.running_var"])

    # ---- 4d. BASELINE configs[0] / SURVEY 8(d) cfg1 AS WRITTEN: torch.manual_seed(0), x1, x2 = randn(64,3,32,32),
    #          resnet18(crop_measures=1), MLP(8192,1024,128), tau = 0.5, one two-view step (h1 detached) ----
    torch.manual_seed(0)
    x1 = torch.randn(64, 3, 32, 32)
    x2 = torch.randn(64, 3, 32, 32)
    m = build_ref("resnet18", 1, 512 * 16, 64, (32, 32), 0.25); m.train()
    with torch.no_grad():
        z1 = m.g(m.f(x1))
    z2 = m.g(m.f(x2))
    z2.retain_grad()
    loss, logits, labels = Objective.contrastive_loss(hidden1=z1.data, hidden2=z2, temperature=0.5)
    loss.backward()
    save("r18_cfg1_literal.npz", z1=z1, z2=z2, loss=loss, dh2=z2.grad, logits=logits,
         g_conv1=m.f.conv1.weight.grad, g_fc2_b=m.g.layers[2].bias.grad,
         gnorms=np.array([p.grad.norm().item() for p in m.parameters()]))

    # ---- 5. host utilities ----
    class A:  # minimal stand-in for the optimizer the schedule reads (Model_Util.py:11-15)
        pass
    rows = []
    for (step, warm, nex, bs, W, ep, scal, base) in [(1, 10, 1000, 64, 8, 190, "linear", 0.01), (100, 10, 1000, 64, 8, 190, "linear", 0.01),
                                                      (156, 10, 1000, 64, 8, 190, "linear", 0.01), (2000, 10, 1000, 64, 8, 190, "sqrt", 0.01),
                                                      (5, 0, 1000, 64, 1, 10, "linear", 0.1), (3000, 10, 1000, 64, 8, 190, "linear", 0.01)]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.0)
        opt.state[p]["step"] = step
        Model_Util.learning_rate_schedule(dict(optimizer=opt, warmup_epochs=warm, num_examples=nex, batch_size=bs, world_size=W,
                                               learning_rate_scaling=scal, base_learning_rate=base, train_epochs=ep))
        rows.append([step, warm, nex, bs, W, ep, 0 if scal == "linear" else 1, base, opt.param_groups[0]["lr"]])
    torch.manual_seed(7)
    preds = torch.randn(32, 20); tgt = torch.randint(0, 20, (32,))
    onehot = torch.nn.functional.one_hot(tgt, 40)
    topk = [Model_Util.top_k_accuracy(preds, tgt, k).item() for k in (1, 5)] + [Model_Util.top_k_accuracy(preds, onehot, k).item() for k in (1, 5)]
    known = [Model_Util.top_k_accuracy(torch.tensor([[.1, .9, 0], [.8, .1, .1]]), torch.tensor([1, 2]), k).item() for k in (1, 2)]
    torch.manual_seed(0)
    z1, z2 = torch.randn(8, 128), torch.randn(8, 128)
    legacy = SimCLR.compute_loss(z1, z2, 0.5)
    save("host_utils.npz", lr_rows=np.array(rows, dtype=np.float64), topk=np.array(topk), topk_known=np.array(known), legacy_loss=legacy)
