#!/usr/bin/env python3
"""SimCLR ResNet-50 training-step benchmark on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one per-GPU batch of synthetic source
images already resident in HBM (reference --num-fixations 1
semantics, Contrastive_Learning.py:638-700): two-view HIP augmentation ->
view-1 forward (no grad, train-mode BN) -> view-2 forward -> NT-Xent (h1
detached; embedding all-gather when N > 1) -> backward -> (gradient all-reduce
when N > 1) -> Adam.  Workload at every N: BASELINE configs[1] per GPU —
ResNet-50 (reference stem: 7x7 stride 1, no max-pool), 3x224x224, per-GPU
batch 256, bf16 storage / fp32 MFMA accumulate, 4x4 adaptive pool + MLP(32768,
1024,128), temperature 0.5 — i.e. weak scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-active-ai_amd")
SIM = os.path.join(PKG, "SimCLR")
for d in (ROOT, PKG, SIM, os.path.join(SIM, "ResNet"), os.path.join(SIM, "MLP")):
    if d not in sys.path:
        sys.path.insert(0, d)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters): dense bf16 MFMA, HBM3E
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
PMC_SUMMARY = "r01_pmc_traffic_b256_v5.json"  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE cfg2: 256)")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--temperature", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--profile-table", default="", help="write the per-kernel table (JSON) here")
    ap.add_argument("--detail", action="store_true", help="per-shape conv rows in the profile table")
    ap.add_argument("--recompute", action="store_true",
                    help="block recompute in the backward (fits --batch 512 = global 4096 on 8 GPUs in 288 GB; one extra forward)")
    return ap.parse_args()


def build(args, device, world):
    import resnet as rn
    import multilayerPerceptron as mlp
    import SimCLR
    import Model_Util
    norm = torch.nn.SyncBatchNorm if world > 1 else torch.nn.BatchNorm2d  # the drivers always pass SyncBatchNorm (Contrastive_Learning.py:240-252)
    torch.manual_seed(1234)  # identical random-init weights on every rank (stands in for DDP's constructor broadcast)
    f = getattr(rn, args.arch)(crop_measures=1, norm_layer=norm)
    exp = 4 if hasattr(f.layer1[0], "conv3") else 1
    g = mlp.MLP(512 * exp * 16, 1024, 128)
    model = SimCLR.SimCLR_Module(f, g, args.batch, (args.img, args.img), device).to(device)
    model.head_pool = 4
    model.train()

    class A:
        optimizer, lr, momentum, weight_decay = "adam", 1e-3, 0.9, 0.0
    return model, Model_Util.get_optimizer(model, A)


def make_step(args, model, opt, device, rank, world):
    import Objective
    from maai_hip import kernels as K
    from maai_hip.dist import GradAllReduce
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    images = torch.randint(0, 256, (args.batch, args.img, args.img, 3), dtype=torch.uint8, device=device, generator=gen)
    sync = GradAllReduce(list(model.parameters())) if world > 1 else None
    state = {"it": 0}

    def step():
        it = state["it"]
        state["it"] += 1
        p1 = K.augment_params(args.batch, args.img, args.img, seed=1000 + rank, view=2 * it, device=device)
        p2 = K.augment_params(args.batch, args.img, args.img, seed=1000 + rank, view=2 * it + 1, device=device)
        v1 = K.augment_view_u8(images, p1, args.img, args.img)
        v2 = K.augment_view_u8(images, p2, args.img, args.img)
        with torch.no_grad():
            h1 = model([v1])
        h2 = model([v2])
        loss, _, _ = Objective.contrastive_loss(hidden1=h1.data, hidden2=h2, temperature=args.temperature,
                                                local_rank=rank, world_size=world, device=device)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if sync is not None:
            sync()
        opt.step()
        return loss
    return step


def cpu_baseline(args):
    """The oracle (a CPU port, fp32 torch-CPU ops) on a bounded sample of the same workload."""
    from oracle import simclr_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))  # a one-GPU box's CPU share is 16 cores
    b = args.cpu_batch
    exp = O.expansion(args.arch)
    sd = O.pattern_state_dict(args.arch, 1, 512 * exp * 16)
    g = torch.Generator().manual_seed(0)
    x1 = torch.randint(0, 256, (b, 3, args.img, args.img), generator=g).float()
    x2 = torch.randint(0, 256, (b, 3, args.img, args.img), generator=g).float()
    opt = {}
    sys.stderr.write("[bench] cpu baseline: %d threads, batch %d ...\n" % (torch.get_num_threads(), b))
    sys.stderr.flush()
    t0 = time.time()
    for i in range(args.cpu_steps):
        O.train_step(sd, opt, x1, x2, args.arch, args.temperature, 1e-3, pool=4)
        sys.stderr.write("[bench] cpu baseline step %d done at %.1f s\n" % (i, time.time() - t0))
        sys.stderr.flush()
    dt = (time.time() - t0) / args.cpu_steps
    return dict(value=round(b / dt, 3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample="%s 3x%dx%d, batch %d, %d steps (fp32, torch-CPU oracle)" % (args.arch, args.img, args.img, b, args.cpu_steps))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # MAAI_BENCH_REHEARSE=1: all ranks share cuda:0 over gloo — a one-GPU rehearsal of the N > 1 code path
    rehearse = os.environ.get("MAAI_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    if args.recompute:
        from maai_hip import engine
        engine.set_recompute(True)
    model, opt = build(args, device, world)
    step = make_step(args, model, opt, device, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    # one extra, untimed, instrumented step: per-kernel HIP-event durations on the launch stream
    from maai_hip import kernels as K
    with K.profile() as prof:
        step()
    table = prof.table()
    if args.detail:
        K.DETAIL[0] = True
        with K.profile() as prof2:
            step()
        K.DETAIL[0] = False
        detail = prof2.table()
        if rank == 0 and args.profile_table:
            with open(args.profile_table + ".detail", "w") as fh:
                json.dump(detail, fh, indent=1)
    roof = None
    if table:
        # Dominant kernel by time.  conv_igemm is a mix of MFMA-bound (3x3) and HBM-bound (1x1) layers: both
        # fractions are computed from the same live HIP-event durations and the roof it sits closer to is reported
        # as the bound; "attainable_frac" is the per-shape roofline time (max of the two roofs, summed over the
        # kernel's shapes, from the per-shape pass below) over the measured time.
        dom = max(table, key=lambda k: table[k]["ms"])
        t = table[dom]
        sec = t["ms"] * 1e-3
        tf = t["flops"] / sec / 1e12
        gbs = t["bytes"] / sec / 1e9
        f_mfma, f_hbm = tf / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
        common = dict(kernel=dom, traffic=None, launches=t["launches"], avg_launch_ms=round(t["ms"] / t["launches"], 4),
                      share_of_step=round(t["ms"] / ms, 3), frac_mfma=round(f_mfma, 4), frac_hbm=round(f_hbm, 4),
                      achieved_TFLOPs=round(tf, 2), achieved_GBs=round(gbs, 1))
        if f_mfma >= f_hbm:
            roof = dict(bound="mfma", achieved=round(tf, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(f_mfma, 4), **common)
        else:
            roof = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(f_hbm, 4), **common)
        K.DETAIL[0] = True
        with K.profile() as prof3:
            step()
        K.DETAIL[0] = False
        rows = [v for k, v in prof3.table().items() if k.split(" ")[0].split("[")[0] == dom and v["ms"] > 0]
        if rows:
            ideal = sum(max(v["flops"] / (PEAK_BF16_TFLOPS * 1e12), v["bytes"] / (PEAK_HBM_GBS * 1e9)) for v in rows)
            roof["attainable_frac"] = round(ideal * 1e3 / sum(v["ms"] for v in rows), 4)
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
    # figure measured for THIS command by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950
    # correction applied) is read from the committed summary when the workload matches; otherwise null.
    if roof is not None and args.arch == "resnet50" and args.img == 224 and args.batch == 256:
        pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if os.path.exists(pmc):
            with open(pmc) as fh:
                k = json.load(fh)["kernels"].get(roof["kernel"])
            if k:
                roof["traffic"] = round(k["hbm_bytes_per_launch"] / 1e9, 3)
                roof["traffic_unit"] = "GB/launch (PMC, profiles/%s)" % PMC_SUMMARY
                roof["algorithmic_GB_per_launch"] = round(table[roof["kernel"]]["bytes"] / table[roof["kernel"]]["launches"] / 1e9, 3)
    if rank == 0 and args.profile_table:
        with open(args.profile_table, "w") as fh:
            json.dump({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in table.items()}, fh, indent=1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            "metric": "images/sec SimCLR ResNet-50 224px, global batch 4096, 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SimCLR %s 3x%dx%d (reference stem), per-GPU batch %d, two-view step: aug + fwd(view1, no grad) + "
                                   "fwd(view2) + NT-Xent tau=%g + bwd + Adam" % (args.arch, args.img, args.img, args.batch, args.temperature),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "loss": float(loss.item()),
                       "recompute": bool(args.recompute), "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if args.arch == "resnet50" and args.img == 224:
            # whole-step view: 4F = 512.1 GFLOP and 2.69 GB of ideal-fusion bf16 conv traffic per source image
            out["step_roofline"] = {"gflop_per_image": 512.1, "ideal_GB_per_image": 2.69,
                                    "frac_mfma": round(512.1e9 * value / world / (PEAK_BF16_TFLOPS * 1e12), 4),
                                    "frac_hbm_ideal_fusion": round(2.69e9 * value / world / (PEAK_HBM_GBS * 1e9), 4)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
