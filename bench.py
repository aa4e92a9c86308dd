#!/usr/bin/env python3
"""SimCLR ResNet-50 training-step benchmark on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started WITHOUT
     a launcher — no WORLD_SIZE in the environment — bench.py starts that launcher itself as a child process, before
     anything touches the GPU, and exits with its return code)

One step = one pass of the hot path over one per-GPU batch of synthetic source
images already resident in HBM (reference --num-fixations 1
semantics, Contrastive_Learning.py:638-700): two-view HIP augmentation ->
view-1 forward (no grad, train-mode BN) -> view-2 forward -> NT-Xent (h1
detached; embedding all-gather when N > 1, view 1's started under the view-2
forward) -> backward (gradient buckets all-reduced on a side stream as the
backward produces them when N > 1) -> Adam.
Workload: ResNet-50 (reference stem: 7x7 stride 1, no max-pool), 3x224x224,
bf16 storage / fp32 MFMA accumulate, 4x4 adaptive pool + MLP(32768,1024,128),
temperature 0.5.  N = 1, 2, 4: BASELINE configs[1] per GPU (per-GPU batch 256,
weak scaling).  N = 8: BASELINE configs[2] = the metric's own configuration,
per-GPU batch 512 (global 4096) with lean activations (the 3x3 convolutions'
normalised inputs are re-formed in the backward instead of stored: 250 GB
allocated / 270 GB reserved of the 309 GB the device reports) and the two
forwards one after the other; the program is fixed by the flags — never by a
free-memory probe — so every rank and every run executes the same launches
(--recompute: block recompute of stage 1 on top, 242 / 264 GB, -6 %;
MAAI_RECOMPUTE_LAYERS=1,2: 201 / 231 GB, -11 %); --batch / --recompute override.
Synthetic images are per-image random low-frequency colour patterns plus
noise, generated on the device, so that the two views of one image correlate
and the contrastive loss is NOT the 2*ln(2N-1) of collapsed embeddings; the
run fails if the loss is not finite or does not move over the timed steps.
Prints ONE JSON line on rank 0.
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
PMC_SUMMARY = "r04_pmc_traffic_b256.json"  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (scripts/profile_round.sh)


def csrc_digest():
    """sha256 over the kernel sources: ties a committed PMC summary to the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fn.encode())
                h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 256 = BASELINE cfg2; 512 at --gpus 8 = cfg3, global 4096)")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--temperature", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU steps; the median is reported (BASELINE.md section 3: >= 10)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed CPU warm-up steps (BASELINE.md section 3: 3)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads (default: every core this process may use)")
    ap.add_argument("--profile-table", default="", help="write the per-kernel table (JSON) here")
    ap.add_argument("--detail", action="store_true", help="per-shape conv rows in the profile table")
    ap.add_argument("--recompute", action="store_true",
                    help="block recompute in the backward (stage 1 unless MAAI_RECOMPUTE_LAYERS names others): less memory, one extra forward of those stages")
    ap.add_argument("--no-recompute", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the two forwards of a step one after the other (default at N = 1 without recompute: the no-grad view on a side "
                         "HIP stream, BatchNorm buffers updated in program order after the join: engine.set_overlap_views)")
    args = ap.parse_args()
    if args.recompute and args.no_recompute:
        ap.error("--recompute and --no-recompute are contradictory")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.batch <= 0:
        args.batch = 512 if args.gpus >= 8 else 256
    # 512 images per GPU at 224 px: the stored activations fit the device (309 GB) only as "lean" ones — the 3x3 convolutions'
    # normalised inputs re-formed in the backward (engine._LEAN) — and with the two forwards one after the other (the side stream's
    # pool would come on top): 250 GB allocated / 270 GB reserved, measured.  --recompute trades 6 % for another 8 GB (stage 1
    # rebuilt in the backward: 242 / 264 GB; MAAI_RECOMPUTE_LAYERS=1,2: 201 / 231 GB).
    args.lean = bool(args.batch * (args.img / 224.0) ** 2 >= 512)
    return args


def self_launch(args):
    """``python3 bench.py --gpus N`` (N > 1) without a launcher: start ``python -m torch.distributed.run`` with the same
    arguments as a CHILD process and return its exit code.  Called before anything in this process has touched the GPU
    (and the GPU is never touched here afterwards: the parent only waits)."""
    import subprocess
    # --standalone: torchrun's own c10d rendezvous on a port IT picks and holds (no bind-and-close race with other launches
    # on a shared box); --local-addr: the container hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


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


def synthetic_images(b, size, device, seed):
    """[b, size, size, 3] u8, generated on the device: per image a sum of three random low-frequency plane waves per
    colour channel (random orientation, wavelength 1/1 .. 1/6 of the image, phase, amplitude) around a random mean
    colour, plus +-12 levels of per-pixel noise.  Two random crops of ONE image share its colours and orientation
    statistics; two different images do not — unlike uniform noise, for which every embedding is the same."""
    gen = torch.Generator(device=device).manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(size, device=device, dtype=torch.float32),
                            torch.arange(size, device=device, dtype=torch.float32), indexing="ij")
    out = torch.empty((b, size, size, 3), dtype=torch.uint8, device=device)
    chunk = 32
    for b0 in range(0, b, chunk):
        n = min(chunk, b - b0)
        img = 40.0 + 175.0 * torch.rand((n, 1, 1, 3), device=device, generator=gen)
        for _ in range(3):
            theta = 3.14159265 * torch.rand((n, 1, 1, 1), device=device, generator=gen)
            freq = 6.2831853 * (1.0 + 5.0 * torch.rand((n, 1, 1, 1), device=device, generator=gen)) / size
            phase = 6.2831853 * torch.rand((n, 1, 1, 3), device=device, generator=gen)
            amp = 50.0 * torch.rand((n, 1, 1, 3), device=device, generator=gen)
            arg = freq * (torch.cos(theta) * xs[None, :, :, None] + torch.sin(theta) * ys[None, :, :, None])
            img = img + amp * torch.sin(arg + phase)
        img = img + 24.0 * (torch.rand((n, size, size, 3), device=device, generator=gen) - 0.5)
        out[b0:b0 + n] = img.clamp_(0.0, 255.0).to(torch.uint8)
    return out


def make_step(args, model, opt, device, rank, world):
    import Objective
    from maai_hip import kernels as K
    from maai_hip import engine
    from maai_hip.dist import GradReducer
    images = synthetic_images(args.batch, args.img, device, 1234 + rank)
    # N > 1: gradient buckets are all-reduced from inside the backward pass (side stream), not after it
    engine.set_grad_hook(GradReducer(list(model.parameters())) if world > 1 else None)
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
        opt.step()
        return loss
    return step


def host_cpu():
    """(model name, physical cores of the host, logical CPUs this process may run on, CPU-time quota of its cgroup or None)"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model == "unknown":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return model, (len(cores) or None), logical, quota


def cpu_baseline(args):
    """The oracle (a CPU port, fp32 torch-CPU ops) on a bounded sample of the same workload, BASELINE.md section 3's protocol:
    ``--cpu-warmup`` (3) untimed steps, then ``--cpu-steps`` (10) timed steps of ``--cpu-batch`` images; the median step time
    is reported, with the CPU model and the cores used."""
    import statistics
    from oracle import simclr_oracle as O
    model, phys, logical, quota = host_cpu()
    # every core this process may use: its affinity mask, capped by its cgroup's CPU-time quota when it has one (more threads
    # than that only fight each other) and by the host's physical cores (SMT siblings do not add fp32 throughput)
    limit = min(logical, phys) if phys else logical
    how = "affinity %d logical CPUs, %s physical cores" % (logical, phys if phys else "?")
    if quota is not None and quota >= 1 and quota < limit:
        limit = int(quota)
        how += ", cgroup quota %.1f CPUs" % quota
    if args.cpu_threads > 0:
        limit = args.cpu_threads
        how += ", --cpu-threads %d" % args.cpu_threads
    torch.set_num_threads(max(1, limit))
    b = args.cpu_batch
    exp = O.expansion(args.arch)
    sd = O.pattern_state_dict(args.arch, 1, 512 * exp * 16)
    imgs = synthetic_images(2 * b, args.img, torch.device("cpu"), 99).permute(0, 3, 1, 2).float().contiguous()
    x1, x2 = imgs[:b], imgs[b:]
    opt = {}
    sys.stderr.write("[bench] cpu baseline: %s, %d threads (%s), batch %d, %d warm-up + %d timed steps ...\n"
                     % (model, torch.get_num_threads(), how, b, args.cpu_warmup, args.cpu_steps))
    sys.stderr.flush()
    times = []
    for i in range(args.cpu_warmup + args.cpu_steps):
        t0 = time.time()
        O.train_step(sd, opt, x1, x2, args.arch, args.temperature, 1e-3, pool=4)
        dt = time.time() - t0
        if i >= args.cpu_warmup:
            times.append(dt)
        sys.stderr.write("[bench] cpu baseline step %d: %.1f s%s\n" % (i, dt, " (warm-up, not counted)" if i < args.cpu_warmup else ""))
        sys.stderr.flush()
    med = statistics.median(times)
    return dict(value=round(b / med, 3), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                cpu_model=model, host_physical_cores=phys, usable_logical_cpus=logical, threads_chosen_from=how,
                sample="%s 3x%dx%d, batch %d, median of %d timed steps after %d warm-up steps (BASELINE.md section 3's protocol; fp32, "
                       "torch-CPU oracle, %d threads); step times min %.1f / median %.1f / max %.1f s"
                       % (args.arch, args.img, args.img, b, args.cpu_steps, args.cpu_warmup, torch.get_num_threads(), min(times), med, max(times)))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (start it as `python3 bench.py --gpus N`, or under "
                         "torch.distributed.run --nproc-per-node N)" % (world, args.gpus))

    rc_layers = None
    if args.recompute:
        from maai_hip import engine
        # Which stages are recomputed is part of the command, not of the machine's state: stage 1 (it holds ~half of the
        # activation bytes; with lean activations — on together with recompute, engine._LEAN — 512 images then take 242 GB
        # allocated / 264 GB reserved, measured; stages 2-4 stay stored and are not run a second time) unless
        # MAAI_RECOMPUTE_LAYERS says otherwise (1,2: 201 / 231 GB, 5 % slower).  Every rank of every run executes the same program.
        rc_layers = sorted(engine._RECOMPUTE["layers"]) if os.environ.get("MAAI_RECOMPUTE_LAYERS") else [1]
        engine.set_recompute(True, rc_layers)
    model, opt = build(args, device, world)
    step = make_step(args, model, opt, device, rank, world)
    from maai_hip import engine as _engine
    if args.lean and os.environ.get("MAAI_LEAN_ACT", "auto") == "auto":
        _engine.set_lean_activations(True)
    overlap = world == 1 and not args.recompute and not args.lean and not args.no_overlap and os.environ.get("MAAI_OVERLAP_VIEWS", "1") != "0"
    _engine.set_overlap_views(overlap)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(args.warmup):
            step()
        barrier()
        losses = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
            losses.append(loss)   # device scalars: read after the timed region
        barrier()
        dt = time.perf_counter() - t0
    except torch.cuda.OutOfMemoryError as e:
        # the program is fixed by the flags (no free-memory probe picks another one behind the user's back): say which flag helps
        hint = ("MAAI_RECOMPUTE_LAYERS=1,2 recomputes stage 2 as well (-~40 GB at 512 images), 1,2,3 stage 3 too" if args.recompute
                else "--recompute rebuilds the blocks of stage 1 in the backward instead of storing their activations (-8 GB at 512 images; "
                     "MAAI_RECOMPUTE_LAYERS=1,2 with it: -49 GB)")
        raise SystemExit("bench.py: out of HBM at --batch %d (recompute %s): %s\n%s" % (args.batch, rc_layers if args.recompute else "off", hint, e))
    losses = [float(v.item()) for v in losses]
    if not all(v == v and abs(v) < 1e30 for v in losses):
        raise SystemExit("bench.py: the loss is not finite over the timed steps: %s" % losses)
    if len(losses) > 1 and max(losses) - min(losses) < 1e-6:
        raise SystemExit("bench.py: the loss does not move over the timed steps (%s): the step is not training" % losses)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    # one extra, untimed, instrumented step: per-kernel HIP-event durations on the launch stream — with the two forwards
    # one after the other (overlapped, a launch's events also span the other stream's launches it shares the device with)
    _engine.set_overlap_views(False)
    from maai_hip import kernels as K
    with K.profile() as prof:
        step()
    table = prof.table()
    ktable = prof.kernel_table()
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
        # roofline.achieved / frac use SURVEY §8(d)'s algorithmic bytes — every convolution reads its input once and
        # writes its output once, nothing else — per launch, over the live HIP-event duration of the launch; the
        # as-built figure (every tensor the launch's epilogue also reads or writes) is kept beside it
        gbs = t["bytes_8d"] / sec / 1e9
        gbs_built = t["bytes"] / sec / 1e9
        f_mfma, f_hbm = tf / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
        common = dict(kernel=dom, traffic=None, launches=t["launches"], avg_launch_ms=round(t["ms"] / t["launches"], 4),
                      share_of_step=round(t["ms"] / ms, 3), frac_mfma=round(f_mfma, 4), frac_hbm=round(f_hbm, 4),
                      achieved_TFLOPs=round(tf, 2), achieved_GBs=round(gbs, 1),
                      algorithmic_GB_per_launch=round(t["bytes_8d"] / t["launches"] / 1e9, 3),
                      as_built_GB_per_launch=round(t["bytes"] / t["launches"] / 1e9, 3), as_built_frac_hbm=round(gbs_built / PEAK_HBM_GBS, 4))
        if f_mfma >= f_hbm:
            roof = dict(bound="mfma", achieved=round(tf, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(f_mfma, 4), **common)
        else:
            roof = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(f_hbm, 4), **common)
        # the five DEVICE kernels that take most of the step, by the names rocprofv3's kernel trace gives them (profiles/
        # r04_bench_b256_kernel_stats.csv: Calls / traced steps = launches, AverageNs = avg_ms): SURVEY 8(d) bytes and flops
        # per launch and the fraction of the roof each sits under, so that the family figure above can be recomputed name by name
        top = sorted(ktable.items(), key=lambda kv: -kv[1]["ms"])[:5]
        rows5 = []
        for kn, v in top:
            ksec = v["ms"] * 1e-3
            fh, fm = v["bytes_8d"] / ksec / 1e9 / PEAK_HBM_GBS, v["flops"] / ksec / 1e12 / PEAK_BF16_TFLOPS
            rows5.append(dict(name=kn, launches=v["launches"],
                              avg_ms=round(v["ms"] / v["launches"], 4), ms_per_step=round(v["ms"], 3),
                              algorithmic_GB_per_launch=round(v["bytes_8d"] / v["launches"] / 1e9, 3),
                              as_built_GB_per_launch=round(v["bytes"] / v["launches"] / 1e9, 3),
                              GFLOP_per_launch=round(v["flops"] / v["launches"] / 1e9, 1),
                              bound="hbm" if fh >= fm else "mfma", frac=round(max(fh, fm), 4), frac_hbm=round(fh, 4), frac_mfma=round(fm, 4)))
        roof["kernels"] = rows5
        K.DETAIL[0] = True
        with K.profile() as prof3:
            step()
        K.DETAIL[0] = False
        rows = [v for k, v in prof3.table().items() if k.split(" ")[0].split("[")[0] == dom and v["ms"] > 0]
        if rows:
            ideal = sum(max(v["flops"] / (PEAK_BF16_TFLOPS * 1e12), v["bytes_8d"] / (PEAK_HBM_GBS * 1e9)) for v in rows)
            roof["attainable_frac"] = round(ideal * 1e3 / sum(v["ms"] for v in rows), 4)
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
    # figure measured for THIS command by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950
    # correction applied; scripts/profile_round.sh) is read from the committed summary — but only while that summary
    # was measured on the kernels being run now (it records a digest of csrc/); otherwise null.
    if roof is not None and args.arch == "resnet50" and args.img == 224 and args.batch == 256:
        pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if os.path.exists(pmc):
            with open(pmc) as fh:
                summary = json.load(fh)
            k = summary.get("kernels", {}).get(roof["kernel"])
            if k and summary.get("csrc_digest") == csrc_digest():
                roof["traffic"] = round(k["hbm_bytes_per_launch"] / 1e9, 3)
                roof["traffic_unit"] = "GB/launch (PMC, profiles/%s)" % PMC_SUMMARY
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(roof["algorithmic_GB_per_launch"], 1e-9), 3)
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
            "config": {"workload": "SimCLR %s 3x%dx%d (reference stem), per-GPU batch %d%s, two-view step: aug + fwd(view1, no grad) + "
                                   "fwd(view2) + NT-Xent tau=%g + bwd + Adam; BASELINE %s"
                                   % (args.arch, args.img, args.img, args.batch, (" with block recompute" if args.recompute else "") + (", lean activations" if _engine._lean() else ""),
                                      args.temperature, "configs[2] (global batch 4096)" if world * args.batch == 4096 else "configs[1] per GPU"),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "loss": losses[-1], "loss_first_timed_step": losses[0],
                       "recompute": bool(args.recompute), "recompute_layers": rc_layers, "overlap_views": bool(overlap), "lean_activations": bool(_engine._lean()), "hbm_total_GB": round(torch.cuda.mem_get_info()[1] / 1e9, 1), "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                       "peak_hbm_reserved_GB": round(torch.cuda.max_memory_reserved() / 1e9, 1)},
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
