"""GPU: the data-parallel step end to end on the real engines.  Two ranks share the one GPU of the
test box and talk over gloo (RCCL needs one GPU per rank; the 8-GPU RCCL run is the driver's
scaling bench); with two or more GPUs visible the *_over_rccl tests run the same workers one GPU per rank over RCCL
(torch.distributed "nccl" and the C-ABI communicator).  What is exercised here is everything above the collective: arena bucketing driven
by the engines' backward callbacks, side-stream weight gradients, averaging, ArenaAdam on the
reduced arena.  Invariant: ranks that start from the same weights and see DIFFERENT shards hold
bit-identical parameters after a step, and those differ from a purely local step."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_env(rank, world, port, mode):
    """mode "gloo": the ranks share GPU 0 and talk over gloo (the one-GPU test box).  "nccl" / "rccl-abi": one GPU per
    rank over RCCL -- through torch.distributed's "nccl" backend, or through the C-ABI communicator (pai_allreduce,
    PAI_COMM=rccl).  Returns the rank's device index."""
    local = 0 if mode == "gloo" else rank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), PAI_DIST_BACKEND="gloo" if mode == "gloo" else "nccl")
    if mode == "rccl-abi":
        os.environ["PAI_COMM"] = "rccl"
    else:
        os.environ.pop("PAI_COMM", None)
    return local


def _worker(rank, world, port, precision, family, q, mode="gloo", grad_dtype=None):
    import sys
    sys.path.insert(0, ROOT)
    local = _rank_env(rank, world, port, mode)
    try:
        import numpy as np
        import torch.distributed as dist
        import pai_bootstrap
        pai = pai_bootstrap.load()
        from thesis_pai_reconstruction_amd import dist as pdist
        pdist.init_from_env()
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        torch.manual_seed(100 + rank)                       # different initial weights per rank ...
        composable = family in ("resnext_unet", "trans_unet")
        if family == "resnext_unet":                        # composable path: gradients outside the arenas (MultiAdam)
            m = pai.ResUnetGAN(1, 1, "next", (1, 2, 2), 0.0, "gan").to(dev)
        elif family == "trans_unet":
            m = pai.TransUnetGAN(1, 1, (1, 1, 1, 2, 2), 2, 0.0, "gan").to(dev)
        else:
            cls = pai.AttentionUnetGAN if family == "attention_unet" else pai.Pix2Pix
            m = cls(1, 1, (1, 2, 2, 4), 0.0, "gan").to(dev)
        m.set_precision(precision)
        m.train()
        pdist.broadcast_parameters(m)                       # ... aligned here
        red = pdist.GradReducer(bucket_bytes=(64 << 10) if composable else (1 << 20), grad_dtype=grad_dtype)
        red.attach(m)
        # what finish() finds already issued: the composable networks' fixed buckets leave from post-accumulate hooks
        # while the backward pass is still running, like the arena buckets of the engines
        at_finish = []
        orig_finish = red.finish

        def counted_finish():
            before = red.stats.get("foreign_buckets", 0)
            orig_finish()
            at_finish.append((before, red.stats.get("foreign_buckets", 0)))
        red.finish = counted_finish
        assert red.plannable()
        assert red.rccl_ranks() == (0 if mode == "gloo" else world), (mode, red.rccl_ranks())
        assert (red.comm is not None) == (mode == "rccl-abi")

        class T:
            reducer = red
            def _log(self, *a): pass
        m.trainer = T()
        rng = np.random.default_rng(7 + rank)               # different shard per rank
        side = 256 if family == "trans_unet" else 64        # (the TransUNet's positional embedding fixes the image size)
        nimg = 2 if family == "trans_unet" else 4
        x = torch.from_numpy(rng.random((nimg, 1, side, side), dtype=np.float32) * 2 - 1).to(dev)
        t = torch.from_numpy(rng.random((nimg, 1, side, side), dtype=np.float32) * 2 - 1).to(dev)
        before = torch.cat([p.detach().reshape(-1).clone() for p in m.parameters()])
        # a twin that takes the ordinary path (x 1/R pass and Adam behind the last bucket) on the same shards
        twin = None
        if not composable:
            import copy
            twin = copy.deepcopy(m)
            red2 = pdist.GradReducer(bucket_bytes=1 << 20, grad_dtype=grad_dtype, comm=red.comm)
            red2.attach(twin)

            class T2:
                reducer = red2
                def _log(self, *a): pass
            twin.trainer = T2()
        os.environ["PAI_NO_STREAM_ADAM"] = "1"
        m.training_step((x, t), 0)             # the first fused step moves the parameters into the arena
        os.environ["PAI_NO_STREAM_ADAM"] = "0"
        os.environ["PAI_DDP_STREAM_ADAM"] = "1"     # the streamed update under a reducer is opt-in (optim.ArenaAdam.arm_streaming)
        if twin is not None:
            twin.training_step((x, t), 0)
            twin.load_state_dict(m.state_dict())
            for oa, ob in zip(m.optimizers(), twin.optimizers()):
                ob.load_state_dict(oa.state_dict())
        m.training_step((x, t), 1)             # streamed: every reduced bucket is averaged + updated on the post stream
        if twin is not None:
            # The twin takes the ordinary path (x 1/R pass and one Adam pass behind the last bucket).  Its averaged
            # gradients must agree with the streamed model's up to the fp32-atomics noise of two backward passes from
            # the same weights (a missed or doubled x 1/R would show); it then STEPS ON THE STREAMED MODEL'S gradients,
            # so that everything downstream -- the generator phase sees the discriminator this step updated, and
            # Adam's first steps turn noise-decided signs into +-lr -- can be compared bit for bit.
            gaps = []
            for oa, ob in zip(m.optimizers(), twin.optimizers()):
                def step(*a, oa=oa, ob=ob, orig=ob.step, **k):
                    ga, gb = oa._engine.arena().flat, ob._engine.arena().flat
                    gaps.append(float((ga.double() - gb.double()).norm()) / max(float(ga.double().norm()), 1e-30))
                    gb.copy_(ga)
                    return orig(*a, **k)
                ob.step = step
            os.environ["PAI_NO_STREAM_ADAM"] = "1"
            twin.training_step((x, t), 1)
            os.environ["PAI_NO_STREAM_ADAM"] = "0"
        torch.cuda.synchronize()
        if twin is not None:
            assert red.stats.get("post_buckets", 0) >= 2 and red2.stats.get("post_buckets", 0) == 0, red.stats
            assert len(gaps) == 2 and max(gaps) <= 1e-3, gaps     # (measured ~1e-6; a wrong 1/R is O(1))
            for (k, p), (_, q2) in zip(m.named_parameters(), twin.named_parameters()):
                assert torch.equal(p.detach(), q2.detach()), k
            for oa, ob in zip(m.optimizers(), twin.optimizers()):
                sa_, sb_ = oa.state_dict()["state"], ob.state_dict()["state"]
                assert len(sa_) == len(sb_) > 0
                for i in sa_:
                    assert float(sa_[i]["step"]) == float(sb_[i]["step"]) == 2
                    for key in ("exp_avg", "exp_avg_sq"):
                        assert torch.equal(sa_[i][key], sb_[i][key]), (i, key)
        after = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        both = [torch.zeros_like(after) for _ in range(world)]
        dist.all_gather(both, after)
        assert torch.equal(both[0], both[1]), "replicas diverged"
        assert float((after - before.cpu()).abs().max()) > 0
        assert red.stats["buckets"] >= 2 * 2 * 2              # several buckets per network per step
        if composable:
            # (entry, exit) counts of the four finish() calls (two backward passes per GAN step; the discriminator's
            # gradients live in an arena): most of the generator's buckets were in flight before finish() ran
            assert len(at_finish) == 4, at_finish
            early = sum(a - (at_finish[i - 1][1] if i else 0) for i, (a, _) in enumerate(at_finish))
            total = red.stats["foreign_buckets"]
            assert total >= 8 and early >= 0.7 * total, (at_finish, red.stats)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def _run_ranks(target, args, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args[0]) + (q,) + tuple(args[1])) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("family", ["pix2pix", "attention_unet", "resnext_unet", "trans_unet"])
@pytest.mark.parametrize("precision", ["32", "bf16-mixed"])
def test_two_rank_step_keeps_replicas_identical(precision, family):
    _run_ranks(_worker, ((precision, family), ()))


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL wants one GPU per rank: needs >= 2 GPUs")


@needs_two_gpus
@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16], ids=["f32buckets", "bf16buckets"])
@pytest.mark.parametrize("mode", ["nccl", "rccl-abi"])
@pytest.mark.parametrize("family", ["pix2pix", "attention_unet", "resnext_unet", "trans_unet"])
def test_two_rank_step_over_rccl(family, mode, grad_dtype):
    """The same invariants with the buckets travelling over RCCL / xGMI, one GPU per rank: torch.distributed's "nccl"
    backend and the C-ABI communicator (PAI_COMM=rccl -> pai_allreduce), fp32 and bf16 buckets, with the streamed
    data-parallel Adam (PAI_DDP_STREAM_ADAM=1, set by the worker for its second step).  Skipped on the one-GPU box; the
    first multi-GPU run of this suite is then a measurement, not a debugging session (reference main.py:123-136: DDP
    comes with pl.Trainer)."""
    _run_ranks(_worker, (("bf16-mixed", family), (mode, grad_dtype)))


@needs_two_gpus
@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16], ids=["f32buckets", "bf16buckets"])
@pytest.mark.parametrize("mode", ["nccl", "rccl-abi"])
def test_reduced_gradients_equal_mean_of_local_over_rccl(mode, grad_dtype):
    _run_ranks(_mean_worker, (("pix2pix",), (mode, grad_dtype)))


def _planned_worker(rank, world, port, family, q, mode="gloo"):
    """Data-parallel steps replayed from launch plans (plan.PlannedStep): the torch.distributed collectives are host
    nodes between C-side segments (gloo / "nccl"), or plan nodes themselves (rccl-abi).  Against an eager twin with a
    reducer of its own on the same shards, step by step from synchronised states: d_loss and the metrics bit for bit,
    parameters to the atomics' noise, and the two ranks bit-identical throughout."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    local = _rank_env(rank, world, port, mode)
    try:
        import copy
        import numpy as np
        import torch.distributed as dist
        import pai_bootstrap
        pai = pai_bootstrap.load()
        from thesis_pai_reconstruction_amd import dist as pdist
        from thesis_pai_reconstruction_amd.plan import PlannedStep
        from _gpu_util import sync_training_state
        pdist.init_from_env()
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        torch.manual_seed(100)
        composable = family == "resnext_unet"      # gradients outside the arenas: the reducer's hook-fed fixed buckets
        if composable:
            eager = pai.ResUnetGAN(1, 1, "next", (1, 2, 2), 0.0, "gan").to(dev)
        else:
            cls = pai.AttentionUnetGAN if family == "attention_unet" else pai.Pix2Pix
            eager = cls(1, 1, (1, 2, 2, 4), 0.0, "gan").to(dev)
        eager.set_precision("bf16-mixed")
        eager.train()
        pdist.broadcast_parameters(eager)
        planned = copy.deepcopy(eager)
        reds = []
        for m in (eager, planned):
            red = pdist.GradReducer(bucket_bytes=(64 << 10) if composable else (1 << 20), comm=reds[0].comm if reds else None)
            red.attach(m)
            assert red.plannable()
            reds.append(red)

            class T:
                reducer = red
                def _log(self, *a): pass
            m.trainer = T()
        ps = PlannedStep(planned, warmup=3)
        steps = 9
        for s in range(steps):
            rng = np.random.default_rng(1000 * rank + s)         # different shard per rank and step
            x = torch.from_numpy(rng.random((4, 1, 64, 64), dtype=np.float32) * 2 - 1).to(dev)
            t = torch.from_numpy(rng.random((4, 1, 64, 64), dtype=np.float32) * 2 - 1).to(dev)
            logs = {}
            for name, m, step in (("e", eager, eager.training_step), ("p", planned, ps)):
                got = {}
                m.trainer._log = lambda k, v, got=got: got.__setitem__(k, v.detach())      # no clone: no torch kernel inside the step
                step((x, t), s)
                logs[name] = got
            torch.cuda.synchronize()
            assert ps.disabled is None, ps.disabled
            for k, v in logs["e"].items():
                a, g = float(v), float(logs["p"][k])
                # (the composable networks' forward passes have fp32 atomics in the BatchNorm statistics of their thin
                #  layers: noise-level differences between any two runs)
                assert (abs(a - g) <= (2e-3 if composable else 1e-4) * max(1.0, abs(a))) if (k == "loss" or composable) else a == g, (s, k, a, g)
            for (k, p), (_, q2) in zip(eager.state_dict().items(), planned.state_dict().items()):
                if not k.endswith("num_batches_tracked"):
                    d = (p.float() - q2.float()).abs()
                    assert float(d.max()) <= (2e-3 if composable else 4.1e-4) and \
                        (d.numel() < 64 or float(d.mean()) <= (1e-4 if composable else 1e-5)), (s, k, float(d.max()))
            after = torch.cat([p.detach().reshape(-1) for p in planned.parameters()]).cpu()
            both = [torch.zeros_like(after) for _ in range(world)]
            dist.all_gather(both, after)
            assert torch.equal(both[0], both[1]), f"replicas diverged at step {s}"
            sync_training_state(eager, planned)
        info = ps.describe()
        assert ps.replays >= 3 and info["plans"] >= 1, info      # (one pack role: under a reducer the update is in step())
        for node in info["nodes"]:
            assert node["launches"] > 50, node
            assert (node["host_nodes"] == 0) == (mode == "rccl-abi"), node      # torch collectives are host nodes
        assert reds[1].stats["buckets"] >= steps * 2 * 2
        if composable:
            assert reds[1].stats["foreign_buckets"] >= steps * 4, reds[1].stats
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("family", ["pix2pix", "attention_unet", "resnext_unet"])
def test_two_rank_planned_step(family):
    _run_ranks(_planned_worker, ((family,), ()))


@needs_two_gpus
@pytest.mark.parametrize("family", ["pix2pix", "resnext_unet"])
@pytest.mark.parametrize("mode", ["nccl", "rccl-abi"])
def test_two_rank_planned_step_over_rccl(mode, family):
    _run_ranks(_planned_worker, ((family,), (mode,)))


def _mean_worker(rank, world, port, family, q, mode="gloo", grad_dtype=None):
    import sys
    sys.path.insert(0, ROOT)
    local = _rank_env(rank, world, port, mode)
    try:
        import numpy as np
        import torch.distributed as dist
        import pai_bootstrap
        pai = pai_bootstrap.load()
        from thesis_pai_reconstruction_amd import dist as pdist
        pdist.init_from_env()
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        cls = pai.AttentionUnetGAN if family == "attention_unet" else pai.Pix2Pix
        rng = np.random.default_rng(17 + rank)
        comm = None
        x = torch.from_numpy(rng.random((4, 1, 64, 64), dtype=np.float32) * 2 - 1).to(dev)
        t = torch.from_numpy(rng.random((4, 1, 64, 64), dtype=np.float32) * 2 - 1).to(dev)
        captured = {}
        for tag in ("local", "dist"):
            torch.manual_seed(5)                           # same weights on both ranks and in both runs
            m = cls(1, 1, (1, 2, 2, 4), 0.0, "gan").to(dev)
            m.set_precision("bf16-mixed")
            m.train()
            if tag == "dist":
                red = pdist.GradReducer(bucket_bytes=256 << 10, grad_dtype=grad_dtype)    # many small buckets: reduced while backward runs
                red.attach(m)
                comm = red.comm

                class T:
                    reducer = red
                    def _log(self, *a): pass
                m.trainer = T()
            for name, net, opt in (("g", m.unet, m.optimizers()[0]), ("d", m.discriminator, m.optimizers()[1])):
                def grab(closure=None, name=name, net=net, tag=tag):
                    torch.cuda.synchronize()
                    captured[(tag, name)] = net.engine.arena().flat.detach().clone().cpu()
                opt.step = grab                             # record the gradient arena instead of updating
            m.training_step((x, t), 0)
        for name in ("g", "d"):
            local = captured[("local", name)]
            both = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(both, local)
            want = (both[0] + both[1]) / 2
            got = captured[("dist", name)]
            assert float((both[0] - both[1]).norm()) > 1e-3 * float(want.norm()), "shards gave identical gradients?"
            # split weight gradients add their partial tiles with float atomics: the order, and so the last bits,
            # differ between two runs of the same step
            err = float((got - want).norm() / want.norm())
            assert err < (1e-4 if grad_dtype in (None, torch.float32) else 1e-2), (name, err)    # bf16 buckets: bf16 rounding of the summands
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("family", ["pix2pix", "attention_unet"])
def test_reduced_gradients_equal_mean_of_local(family):
    """What the optimizer sees under the reducer (buckets all-reduced on the side streams while the backward pass
    is still running) is the mean of the ranks' local gradients -- a bucket launched before its range of the arena
    was final would show up here as an O(1) error."""
    _run_ranks(_mean_worker, ((family,), ()))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two worker ranks before it
    touches the GPU (here both on the one GPU of the box, over gloo), rank 0 prints the ONE JSON line with n_gpus = 2,
    for the weak (64 images per rank in production, 4 here) and the strong (--global-batch) form.  reference
    main.py:123-136 gets its ranks from pl.Trainer."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for extra, scaling, per_gpu in ((["--batch", "4"], "weak", 4), (["--global-batch", "8"], "strong", 4)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline", "--no-kernel-events"] + extra, env=env, capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and rec["config"]["per_gpu_batch"] == per_gpu
        assert rec["config"]["global_batch"] == 8 and rec["config"]["parallelism"] == "dp2" and rec["value"] > 0
