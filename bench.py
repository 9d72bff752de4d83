"""Headline benchmark: train images/s of the Pix2Pix GAN step (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = UnetWrapper.training_step on one synthetic batch already resident in HBM:
D phase + G phase, both Adam updates and the per-step SSIM/PSNR/RMSE (reference
models/wrapper.py:117-162), 256x256x1 pairs, 64 images per GPU (weak scaling), bf16 storage /
fp32 accumulate.  Rank 0 prints ONE JSON line.

roofline: every convolution launch is timed with HIP events recorded on the stream it is launched
on and keyed by the kernel symbol rocprofv3 reports (pai_conv_kernel_name); the kernel with the
largest summed time is the dominant one: achieved = algorithmic FLOPs of its launches / their
summed duration, against the dense bf16 MFMA peak.
cpu_baseline: the oracle's CPU restatement of the same step (same ATen ops the reference
dispatches to), timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MULTS = (1, 2, 4, 8, 8, 8, 8, 8)
TRANS_MULTS = (1, 2, 2, 4, 4)       # TransUnetGAN's class default; the CLI default of 8 levels leaves no patches
SIZE = 256
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E, same guide
# conv MACs per image: generator G, discriminator D, first layers G1, D1
G_MAC, D_MAC, G1_MAC, D1_MAC = 5_947_523_072, 1_646_010_368, 16_777_216, 33_554_432


def gate_mac_per_image() -> int:
    """Forward MACs of the 7 attention gates of BASELINE configs[2]: two C -> C/2
    pointwise convolutions and the C/2 -> 1 head per gated level."""
    total = 0
    for lvl, mult in enumerate(MULTS[:-1]):
        c, sp = mult * 64, SIZE >> (lvl + 1)
        total += sp * sp * (2 * c * (c // 2) + c // 2)
    return total


def step_gflop_per_image(reuse_forward: bool, attention: bool = False) -> float:
    g_passes = 3 if reuse_forward else 4
    g_mac = G_MAC + (gate_mac_per_image() if attention else 0)
    mac = g_passes * g_mac - G1_MAC + 8 * D_MAC - 2 * D1_MAC
    return 2 * mac / 1e9


def host_cores() -> int:
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(attention=False, b4=(3, 20), b64=(3, 5)):
    """The oracle's CPU restatement of the same GAN step (fp32, the ATen kernels the reference dispatches to) on this
    box's host cores, at BOTH batch sizes BASELINE.md section 4 asks for, each with >= 3 warm-up and >= 5 timed steps:
    batch 4 (BASELINE configs[0], the headline `value`: 3 + 20 steps, ~10 s) and batch 64 (the GPU workload's batch:
    3 + 5 steps of ~8 s each), bounded so that the default run still ends within minutes."""
    import oracle
    torch.set_num_threads(host_cores())
    make = oracle.make_attention_unet_state if attention else oracle.make_unet_state

    def run(batch, warmup, steps):
        g = oracle.init_state_portable(make(1, 1, MULTS), 1)
        d = oracle.init_state_portable(oracle.make_disc_state(1), 2)
        rng = np.random.default_rng(1234)
        x = torch.from_numpy(rng.random((batch, 1, SIZE, SIZE), dtype=np.float32) * 2 - 1)
        t = torch.from_numpy(rng.random((batch, 1, SIZE, SIZE), dtype=np.float32) * 2 - 1)
        og, od = oracle.AdamState(), oracle.AdamState()
        for _ in range(warmup):
            oracle.gan_training_step(g, d, og, od, x, t)
        t0 = time.perf_counter()
        for _ in range(steps):
            oracle.gan_training_step(g, d, og, od, x, t)
        return batch * steps / (time.perf_counter() - t0)

    v4 = run(4, *b4)
    v64 = run(64, *b64) if b64 and b64[1] > 0 else None
    return {"value": round(v4, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "physical_cores_visible": os.cpu_count(),
            "sample": f"{b4[1]} fp32 GAN steps at batch 4 (BASELINE configs[0]) after {b4[0]} warm-up steps, oracle/step_ref.py "
                      f"on torch-CPU; value_b64: {b64[1]} timed steps at batch 64 after {b64[0]} warm-up steps, same cores",
            "value_b64": None if v64 is None else round(v64, 3)}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N worker processes (one rank each, torchrun-style
    environment, rendezvous on 127.0.0.1) and wait for them.  This parent never touches the GPU -- it runs before
    any HIP call and only counts devices -- and nothing is re-exec'ed: the workers are ordinary children whose exit
    codes it returns.  On a box with fewer GPUs than ranks (the one-GPU development box) the ranks share device 0
    and talk over gloo (RCCL wants one GPU per rank); rank 0's single JSON line says which."""
    import socket
    import subprocess
    ngpu = torch.cuda.device_count()          # no context is created by counting
    if ngpu < 1:
        raise SystemExit("bench.py needs a GPU: there is no CPU path")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PAI_BENCH_SPAWNED="1")
    if ngpu < n:
        env.setdefault("PAI_DIST_BACKEND", "gloo")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r % ngpu))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# (per-GPU batch, image size) of the BASELINE.json configuration each model family is quoted on
BASELINE_CONFIG = {"pix2pix": (1, 64, 256), "attention_unet": (2, 64, 256), "resnext_unet": (3, 16, 512), "trans_unet": (4, 32, 256)}


def workload_name(args):
    k, b, size = BASELINE_CONFIG[args.model]
    what = {"pix2pix": "Pix2Pix generator+PatchGAN GAN step",
            "attention_unet": "Attention U-Net generator+PatchGAN GAN step",
            "resnext_unet": "Residual U-Net (ResNeXt blocks) generator+PatchGAN GAN step",
            "trans_unet": f"TransUNet (ViT bottleneck, patch size {args.patch_size}) generator+PatchGAN GAN step"}[args.model]
    is_cfg = args.batch == b and args.size == size and "bf16" in args.precision and (args.model != "trans_unet" or args.patch_size == 4)
    tag = f" (BASELINE configs[{k}])" if is_cfg else f" (NOT a BASELINE configuration: configs[{k}] is {b} images/GPU at {size}x{size}, bf16)"
    return f"{what}, {args.size}x{args.size}x1 pairs, {args.batch} images/GPU{tag}"



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling: fixed as N grows)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: total images per step, split evenly over the ranks (overrides --batch; the "
                         "JSON line then says scaling = strong)")
    ap.add_argument("--launch", default="plan", choices=["plan", "eager"],
                    help="plan (default): the step is recorded once into a C-side launch plan (plan.PlannedStep -> "
                         "pai_plan_run) and replayed with ONE C call per step -- same kernels, arguments, streams and "
                         "order as the eager step; eager: every launch issued from Python (PAI_PLAN=0 does the same)")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="wire format of the gradient buckets (dist.GradReducer)")
    ap.add_argument("--bucket-mb", type=float, default=None, help="gradient bucket size in MB (default 32, or PAI_DDP_BUCKET_MB)")
    ap.add_argument("--rccl-algo", default=None, choices=["Ring", "Tree"], help="force NCCL_ALGO (default: RCCL's own choice)")
    ap.add_argument("--rccl-proto", default=None, choices=["Simple", "LL", "LL128"], help="force NCCL_PROTO")
    ap.add_argument("--rccl-channels", type=int, default=None, help="force NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS")
    ap.add_argument("--precision", default="bf16-mixed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket the convolution launches with HIP events in the timed region (no roofline)")
    ap.add_argument("--no-reuse", action="store_true", help="literal two generator forwards per step")
    ap.add_argument("--model", default="pix2pix", choices=["pix2pix", "attention_unet", "resnext_unet", "trans_unet"],
                    help="pix2pix = BASELINE configs[1] (the headline metric); attention_unet = configs[2]; "
                         "resnext_unet = configs[3] (use --size 512 --batch 16); trans_unet = configs[4] (use --batch 32; "
                         "channel_mults pinned to 1,2,2,4,4, SURVEY Q16)")
    ap.add_argument("--patch-size", type=int, default=4,
                    help="trans_unet: ViT patch size (4 = what the reference's main.py passes: d_model 4096, 1.03 B "
                         "parameters; 2 = the class default: d_model 1024, 105 M)")
    ap.add_argument("--size", type=int, default=SIZE, help="image size (configs[3] is quoted at 512)")
    ap.add_argument("--set", dest="tunables", default="",
                    help="name=value,... launch-configuration switches (pai_set_tunable) for A/B runs on one box; the "
                         "line then carries them under config.tunables")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))        # before anything touches the GPU

    import pai_bootstrap
    pai = pai_bootstrap.load()
    from thesis_pai_reconstruction_amd import dist as pdist, ops

    if args.rccl_algo or args.rccl_proto or args.rccl_channels:
        pdist.configure_rccl(algo=args.rccl_algo, proto=args.rccl_proto, min_channels=args.rccl_channels, max_channels=args.rccl_channels)
    rank, local, world = pdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    tunables = {}
    for kv in filter(None, args.tunables.split(",")):
        k, v = kv.split("=")
        tunables[k] = int(v)
        pai.lib.check(pai.lib.load().pai_set_tunable(k.encode(), int(v)), "pai_set_tunable")
    scaling = "weak"
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of {world} ranks")
        args.batch, scaling = args.global_batch // world, "strong"

    torch.manual_seed(0)
    mults = TRANS_MULTS if args.model == "trans_unet" else MULTS
    if args.model == "resnext_unet":
        model = pai.ResUnetGAN(1, 1, "next", MULTS, 0.0, "gan")
    elif args.model == "trans_unet":
        model = pai.TransUnetGAN(1, 1, TRANS_MULTS, args.patch_size, 0.0, "gan")
    else:
        model = (pai.AttentionUnetGAN if args.model == "attention_unet" else pai.Pix2Pix)(1, 1, MULTS, 0.0, "gan")
    model.to(dev)
    model.set_precision(args.precision)
    model.reuse_generator_forward = not args.no_reuse
    model.train()
    model.optimizers()
    reducer = None
    if world > 1:
        pdist.broadcast_parameters(model)
        reducer = pdist.GradReducer(grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32,
                                    bucket_bytes=None if args.bucket_mb is None else int(args.bucket_mb * (1 << 20)))
        reducer.attach(model)

        class _T:  # the hooks UnetWrapper needs from a trainer
            pass
        tr = _T()
        tr.reducer = reducer
        tr._log = lambda name, value: None
        model.trainer = tr

    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.random((args.batch, 1, args.size, args.size), dtype=np.float32) * 2 - 1).to(dev)
    t = torch.from_numpy(rng.random((args.batch, 1, args.size, args.size), dtype=np.float32) * 2 - 1).to(dev)
    batch = (x, t)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    from thesis_pai_reconstruction_amd import plan as pplan
    planned = None
    if args.launch == "plan" and pplan.enabled_by_default():
        planned = pplan.PlannedStep(model, warmup=3)

    def run_step(b, i):
        if planned is not None:
            planned(b, i)
        else:
            model.training_step(b, i)

    # Clock ramp: a box that has been idle starts in a low-power state and takes a few seconds of load
    # to reach its sustained clocks (first 25 steps measured 20-25 % slower than the next 25).  Untimed
    # groups of 10 steps run until two consecutive groups agree within 2 % (at least 2 s, at most 20 s);
    # all ranks take the same decision.  These come BEFORE the W warm-up steps and the K timed ones.
    prewarm_steps, prev, t_start = 0, None, time.perf_counter()
    while True:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for i in range(10):
            run_step(batch, i)
        torch.cuda.synchronize()
        cur = time.perf_counter() - tg
        prewarm_steps += 10
        elapsed = time.perf_counter() - t_start
        stable = prev is not None and abs(cur - prev) <= 0.02 * prev and elapsed >= 2.0
        if os.environ.get("PAI_BENCH_DEBUG"):
            print(f"[clock ramp] t={elapsed:.2f}s {cur * 100:.3f} ms/step", file=sys.stderr, flush=True)
        prev = cur
        flag = torch.tensor([0.0 if (stable or elapsed >= 20.0) else 1.0], device=dev)
        if world > 1:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if float(flag) == 0.0:
            break
    for i in range(args.warmup):
        run_step(batch, i)
    torch.cuda.synchronize()
    # host-side cost of issuing one step (launches + autograd plumbing, or one graph launch), measured without
    # waiting for the GPU: if this approaches ms_per_step the run is launch-bound, not kernel-bound
    tq = time.perf_counter()
    for i in range(3):
        run_step(batch, i)
    host_issue_ms = (time.perf_counter() - tq) / 3 * 1e3
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(batch, i)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    # Per-launch HIP events (two per convolution launch, on the launching stream) for the roofline are
    # taken over a REPEAT of the same K steps, not inside the timed ones: ~150 event records per step
    # put a barrier packet between back-to-back kernels and cost 2-5 ms/step (11.7-15.1 vs 9.9 ms
    # measured), which would make `value` a measurement of the instrumentation.
    prof, prof_hbm = [], []
    if not args.no_kernel_events:
        ops.PROFILE, ops.PROFILE_HBM = [], []
        for i in range(args.steps):
            model.training_step(batch, i)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        prof_hbm, ops.PROFILE_HBM = ops.PROFILE_HBM, None
        barrier()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)

    # ---- per-kernel roofline from the launches' own HIP events --------------------------------------
    def roofline_of(prof, nsteps, dom=None):
        fam = {}
        for kid, op, flops, e0, e1 in prof:
            ms = e0.elapsed_time(e1)
            f = fam.setdefault(kid, {"ms": 0.0, "flops": 0, "launches": 0})
            f["ms"] += ms
            f["flops"] += flops
            f["launches"] += 1
        if not fam:
            return None, None
        if dom is None:
            dom = max(fam, key=lambda k: fam[k]["ms"])
        f = fam[dom]
        achieved = f["flops"] / (f["ms"] * 1e-3) / 1e12
        traffic, traffic_source = None, None
        try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/): they were taken on the DEFAULT
            # workload (BASELINE configs[1]); a kernel of that name in another family's step runs other layers
            if args.model == "pix2pix" and args.batch == 64:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))[dom]["hbm_bytes_per_launch"]
                traffic_source = ("profiles/pmc_traffic_latest.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "workload on the builder's GPU box (scripts/profile_round.sh), NOT measured by this run")
        except (OSError, KeyError, ValueError):
            pass
        return dom, {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": dom,
                     "launches_per_step": f["launches"] / nsteps,
                     "avg_launch_us": round(1e3 * f["ms"] / f["launches"], 2),
                     "gflop_per_launch": round(f["flops"] / f["launches"] / 1e9, 3),
                     "kernel_ms_per_step": {k: round(v["ms"] / nsteps, 3) for k, v in sorted(fam.items())},
                     # every matrix-core kernel family of the step against the same peak: launches, time and algorithmic
                     # FLOPs per step, fraction of the dense bf16 peak
                     "per_kernel": {k: {"launches_per_step": round(v["launches"] / nsteps, 1),
                                        "ms_per_step": round(v["ms"] / nsteps, 3),
                                        "gflop_per_step": round(v["flops"] / nsteps / 1e9, 1),
                                        "frac": round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4)}
                                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}

    def roofline_hbm_of(prof_hbm, nsteps):
        """The HBM-bound passes issued through ops (BatchNorm apply / backward passes of the composable networks, Adam):
        algorithmic bytes over HIP-event time of the family that takes the most time."""
        fam = {}
        for name, nbytes, e0, e1 in prof_hbm:
            f = fam.setdefault(name, {"ms": 0.0, "bytes": 0, "launches": 0})
            f["ms"] += e0.elapsed_time(e1)
            f["bytes"] += nbytes
            f["launches"] += 1
        if not fam:
            return None
        top = max(fam, key=lambda k: fam[k]["ms"])
        f = fam[top]
        gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "traffic": None, "kernel": top, "launches_per_step": f["launches"] / nsteps,
                "ms_per_step": {k: round(v["ms"] / nsteps, 3) for k, v in sorted(fam.items())},
                "algorithmic_gb_per_step": {k: round(v["bytes"] / nsteps / 1e9, 3) for k, v in sorted(fam.items())},
                "note": "co-scheduled with the matrix kernels where the step overlaps them; bytes are algorithmic (tensor sizes)"}

    dom, roofline = roofline_of(prof or [], args.steps)
    roofline_hbm = roofline_hbm_of(prof_hbm or [], args.steps)
    # The timed region co-schedules weight-gradient kernels with the input-gradient chain on a second
    # stream, which stretches every individual launch.  For the kernel's own efficiency the same
    # launches are timed once more with that overlap switched off (3 extra steps, not part of `value`).
    roofline_isolated = None
    engines = [m.engine for m in (model.unet, model.discriminator) if hasattr(type(m), "engine")]
    if any(e._side.on for e in engines) and not args.no_kernel_events:
        saved = [e._side.on for e in engines]
        for e in engines:
            e._side.on = False
        model.training_step(batch, 0)
        torch.cuda.synchronize()
        ops.PROFILE = []
        for i in range(3):
            model.training_step(batch, i)
        torch.cuda.synchronize()
        prof2, ops.PROFILE = ops.PROFILE, None
        for e, on in zip(engines, saved):
            e._side.on = on
        _, roofline_isolated = roofline_of(prof2, 3, dom)
        if roofline is not None:
            roofline["note"] = ("each launch's own HIP start/stop events (hipExtLaunchKernel through pai_profile_arm: the duration "
                                "rocprofv3 reports, no marker packets) over an eager repeat of the K timed steps, outside the timed "
                                "region; weight-gradient kernels co-scheduled on a second stream")
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    reuse = model._can_reuse_forward()
    gflop = step_gflop_per_image(reuse, args.model == "attention_unet")
    gflop_src = "SURVEY 8(d) MAC budget"
    if args.model in ("resnext_unet", "trans_unet"):
        # SURVEY 8(d) gives the MAC budget of configs[1]/[2] only: for the composable families the figure is what the step
        # EXECUTED -- ops.conv_flops (2 x M x Cout x taps x Cin of the dense product, grouped convolutions counted with
        # their block-diagonal zeros removed) summed over the convolution-family launches of the profiled steps
        gflop = (sum(p[2] for p in prof) / args.steps / args.batch / 1e9) if prof else float("nan")
        gflop_src = "sum of ops.conv_flops over the executed convolution-family launches (incl. nn.Linear as 1 x 1)"
    out = {
        "metric": (f"train images/sec ({args.size}x{args.size}, bs={args.batch}) Pix2Pix step" if args.model == "pix2pix" else
                   f"train images/sec ({args.size}x{args.size}, bs={args.batch}) {args.model} GAN step"),
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "bf16" if "bf16" in args.precision else "f32",
        "data": "synthetic",
        # the workload as RUN (batch and size from the flags); the "(BASELINE configs[k])" tag only when they are that
        # configuration's own -- a --batch 8 line must not read as the configs[1] workload
        "config": {"workload": workload_name(args),
                   **({"tunables": tunables} if tunables else {}),
                   "global_batch": world * args.batch, "per_gpu_batch": args.batch,
                   "channel_mults": list(mults), "loss_type": "gan",
                   "generator_forwards_per_step": 1 if reuse else 2, "parallelism": f"dp{world}",
                   "grad_bucket_dtype": args.grad_dtype,
                   **({"backend": torch.distributed.get_backend(), "gpus_visible": torch.cuda.device_count(),
                       "rccl_ranks": reducer.rccl_ranks() if reducer is not None else 0,
                       "gradient_exchange": reducer.describe() if reducer is not None else None} if world > 1 else {})},
        "host_issue_ms_per_step": round(host_issue_ms, 3),
        "launch_mode": ("launch plan (pai_plan_run: one C call per step)" if planned is not None and planned.replays > 0 else
                        "eager" + (f" (plan refused: {planned.disabled})" if planned is not None and planned.disabled else "")),
        "launch_plan": planned.describe() if planned is not None else None,
        "clock_ramp_steps": prewarm_steps,
        "step_conv_gflop_per_image": None if gflop != gflop else round(gflop, 2),
        "step_conv_gflop_source": gflop_src,
        "step_mfma_frac": None if gflop != gflop else round(gflop * 1e9 * value / world / 1e12 / PEAK_BF16_TFLOPS, 4),
        "roofline": roofline,
        "roofline_isolated": roofline_isolated,
        "roofline_hbm": roofline_hbm,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(attention=args.model == "attention_unet") if args.model not in ("resnext_unet", "trans_unet") else None
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
