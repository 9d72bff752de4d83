"""The training step replayed from a C-side launch plan (plan.PlannedStep -> pai_plan_run, include/pai_hip.h "Launch
plans") against the eager step: same losses / metrics, same parameters, same step counts and BatchNorm bookkeeping,
one C call of host time per step (reference step: models/wrapper.py:117-162, one Python call)."""
import time

import numpy as np
import pytest
import torch

import oracle
from oracle.gen_golden import synth_batch
from _gpu_util import sync_training_state as _sync_training_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(pai, family, mults, seed, dtype):
    if family == "pix2pix":
        m = pai.Pix2Pix(1, 1, tuple(mults), 0.0, "gan")
        m.unet.load_state_dict(oracle.init_state_portable(oracle.make_unet_state(1, 1, tuple(mults)), seed, perturb_bn=True))
    else:
        m = pai.AttentionUnetGAN(1, 1, tuple(mults), 0.0, "gan")
        m.unet.load_state_dict(oracle.init_state_portable(oracle.make_attention_unet_state(1, 1, tuple(mults)), seed,
                                                          perturb_bn=True))
    m.discriminator.load_state_dict(oracle.init_state_portable(oracle.make_disc_state(1), seed + 1))
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("family", ["pix2pix", "attention_unet"])
def test_planned_step_matches_eager(pai, family, dtype):
    """Every step starts from the SAME state in both models (the eager model's, copied in place after each comparison:
    two free-running eager models drift apart through the fp32 atomics of the bias / thin-layer gradients, see
    test_gpu_graph.py).  The forward pass has no atomics, so d_loss and the metrics must agree to the last bit
    (the generator's loss sits behind the discriminator's update); parameters after the step agree to what the atomics'
    summation order leaves."""
    from thesis_pai_reconstruction_amd.plan import PlannedStep
    mults, n, size, steps = (1, 2, 4, 8), 4, 64, 10
    batches = [tuple(t.to(DEV) for t in synth_batch(100 + s, n, size)) for s in range(steps)]
    eager, planned = (_build(pai, family, mults, 3, dtype) for _ in range(2))
    ps = PlannedStep(planned, warmup=3)
    for s, b in enumerate(batches):
        eager.logged, planned.logged = {}, {}
        eager.training_step(b, s)
        ps(b, s)
        torch.cuda.synchronize()
        assert ps.disabled is None, ps.disabled
        assert set(eager.logged) == set(planned.logged) == {"d_loss", "loss", "train_ssim", "train_psnr", "train_rmse"}
        for k, v in eager.logged.items():
            a, g = float(v), float(planned.logged[k])
            if k == "loss":     # behind the discriminator's update, whose bias / thin-layer gradients come out of fp32 atomics
                assert abs(a - g) <= 1e-4 * max(1.0, abs(a)), (s, k, a, g)
            else:
                assert a == g, (s, k, a, g)
        for (k, p), (_, q) in zip(eager.state_dict().items(), planned.state_dict().items()):
            if k.endswith("num_batches_tracked"):
                assert int(p) == int(q) == 2 * (s + 1), k
            else:
                # one step apart from identical states: +-lr per element at most where the atomics' summation order flips
                # a sign that Adam's first steps normalise (lr = 2e-4), far less in the mean of a tensor of any size
                d = (p.float() - q.float()).abs()
                assert float(d.max()) <= 4.1e-4 and (d.numel() < 64 or float(d.mean()) <= 1e-5), \
                    (s, k, float(d.max()), float(d.mean()))
        _sync_training_state(eager, planned)
    # steps 1-3 eager warm-up, 4 and 5 recorded (the two roles of the double-buffered packs), 6.. replayed
    assert ps.records == (2 if dtype == torch.bfloat16 else 1), ps.describe()
    assert ps.replays == steps - 3 - ps.records, ps.describe()
    for info in ps.describe()["nodes"]:
        assert info["launches"] > 50 and info["streams"] >= 3 and info["waits"] >= 8, info
    assert planned._pai_opt_steps == eager._pai_opt_steps == 2 * steps
    for oe, og in zip(eager._all_optimizers(), planned._all_optimizers()):
        assert og.total_steps == oe.total_steps == steps
        se, sg = oe.state_dict()["state"], og.state_dict()["state"]
        assert int(next(iter(sg.values()))["step"]) == int(next(iter(se.values()))["step"]) == steps


def test_planned_and_eager_steps_interleave(pai):
    """An eager step between replays (a ragged last batch, a validation pass) flips the pack roles and moves the step
    count: the next planned call finds the plan of the other role (or records one) and the Adam launches replay with the
    right bias correction.  Checked against an eager twin, step by step from synchronised states."""
    from thesis_pai_reconstruction_amd.plan import PlannedStep
    mults, n, size = (1, 2, 4, 8), 4, 64
    eager, planned = (_build(pai, "pix2pix", mults, 11, torch.bfloat16) for _ in range(2))
    ps = PlannedStep(planned, warmup=3)
    ragged = tuple(t.to(DEV) for t in synth_batch(999, 2, size))
    schedule = ["p"] * 6 + ["e", "p", "p", "r", "p", "e", "e", "p", "p"]
    for s, kind in enumerate(schedule):
        b = ragged if kind == "r" else tuple(t.to(DEV) for t in synth_batch(500 + s, n, size))
        eager.logged, planned.logged = {}, {}
        eager.training_step(b, s)
        if kind == "e":
            planned.training_step(b, s)
        else:
            ps(b, s)
        torch.cuda.synchronize()
        assert ps.disabled is None, ps.disabled
        for k, v in eager.logged.items():
            a, g = float(v), float(planned.logged[k])
            assert (abs(a - g) <= 1e-4 * max(1.0, abs(a))) if k == "loss" else a == g, (s, kind, k, a, g)
        for (k, p), (_, q) in zip(eager.state_dict().items(), planned.state_dict().items()):
            if not k.endswith("num_batches_tracked"):
                d = (p.float() - q.float()).abs()
                assert float(d.max()) <= 4.1e-4 and float(d.mean()) <= 1e-5, (s, kind, k, float(d.max()))
        _sync_training_state(eager, planned)
    assert ps.replays >= 5 and len(ps.plans) >= 3, ps.describe()       # two roles at batch 4 + the ragged shape
    for oe, og in zip(eager._all_optimizers(), planned._all_optimizers()):
        assert og.total_steps == oe.total_steps == len(schedule)


def test_plan_replay_is_one_c_call_of_host_time(pai):
    from thesis_pai_reconstruction_amd.plan import PlannedStep
    m = _build(pai, "pix2pix", (1, 2, 4, 8, 8), 5, torch.bfloat16)
    b = tuple(t.to(DEV) for t in synth_batch(7, 8, 128))
    ps = PlannedStep(m, warmup=3)
    for s in range(8):
        ps(b, s)
    torch.cuda.synchronize()
    assert ps.disabled is None and ps.replays >= 2, ps.describe()
    t0 = time.perf_counter()
    for s in range(20):
        ps(b, s)
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(20):
        m.training_step(b, s)
    eager = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"host issue per step: plan {host * 1e3:.3f} ms, eager {eager * 1e3:.3f} ms, {ps.describe()['nodes']}")
    # ~0.75-0.95 ms on an idle box (160 launches + 25 edges at ~5 us); the bound leaves room for a loaded host
    assert host < 2.5e-3 and host < 0.5 * eager
    assert np.isfinite(float(m.logged["loss"]))


def _build_composable(pai, family, dtype):
    torch.manual_seed(0)
    m = {"resnext_unet": lambda: pai.ResUnetGAN(1, 1, "next", (1, 2), 0.0, "gan"),
         "res18_unet": lambda: pai.ResUnetGAN(1, 1, "18", (1, 2), 0.0, "gan"),
         "trans_unet": lambda: pai.TransUnetGAN(1, 1, (1, 2), 2, 0.0, "gan")}[family]()
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("family", ["resnext_unet", "res18_unet", "trans_unet"])
def test_composable_families_replay_from_a_plan(pai, family, dtype):
    """The op-level networks (nnops.py): every launch of their step is a library launch too (gradient fan-in through
    nnops.Fork, filter layout / patch rearrangement kernels, MultiAdam), so the step records and replays like the Pix2Pix
    one.  Same protocol as test_planned_step_matches_eager: both models start every step from the same state."""
    from thesis_pai_reconstruction_amd.plan import PlannedStep
    steps = 8
    size = 256 if family == "trans_unet" else 64
    eager, planned = _build_composable(pai, family, dtype), _build_composable(pai, family, dtype)
    planned.load_state_dict(eager.state_dict())
    ps = PlannedStep(planned, warmup=3)
    for s in range(steps):
        b = tuple(t.to(DEV) for t in synth_batch(300 + s, 2, size))
        eager.logged, planned.logged = {}, {}
        eager.training_step(b, s)
        ps(b, s)
        torch.cuda.synchronize()
        assert ps.disabled is None, ps.disabled
        assert set(eager.logged) == set(planned.logged) == {"d_loss", "loss", "train_ssim", "train_psnr", "train_rmse"}
        for k, v in eager.logged.items():
            a, g = float(v), float(planned.logged[k])
            tol = 1e-4 if k == "loss" else 0.0
            assert abs(a - g) <= tol * max(1.0, abs(a)), (s, k, a, g)
        for (k, p), (_, q) in zip(eager.state_dict().items(), planned.state_dict().items()):
            if k.endswith("num_batches_tracked"):
                assert int(p) == int(q), k
            else:
                d = (p.float() - q.float()).abs()
                # (a conv bias in front of a BatchNorm-led block has a gradient that is fp32 rounding noise: Adam turns its
                #  sign into +-lr steps -- small tensors of that kind are held to the max bound only)
                assert float(d.max()) <= 4.1e-4 and (d.numel() < 1024 or float(d.mean()) <= 1e-5), \
                    (s, k, float(d.max()), float(d.mean()))
        # gradients of the replayed step are where the host expects them
        gp = [p.grad for p in planned.unet.parameters() if p.grad is not None]
        ge = [p.grad for p in eager.unet.parameters() if p.grad is not None]
        assert len(gp) == len(ge) > 10
        scale = max(float(g.norm()) for g in ge)
        # (behind the discriminator's update, whose atomics-ordered bias gradients differ in the last bits: bf16 storage
        #  rounds those differences up to ~1 % at the far end of the generator's backward pass)
        # (a bias in front of a BatchNorm-led block has a gradient that is rounding noise around zero -- in bf16 storage a
        #  few 1e-3 of the largest gradient: absolute floor)
        rtol, floor = (1e-3, 1e-5) if dtype == torch.float32 else (5e-2, 3e-3)
        for a, g in zip(gp, ge):
            assert float((a - g).norm()) <= rtol * float(g.norm()) + floor * scale
        _sync_training_state(eager, planned)
    assert ps.records >= 1 and ps.replays == steps - 3 - ps.records and ps.replays >= 3, ps.describe()
    for info in ps.describe()["nodes"]:
        assert info["launches"] > 100 and info["streams"] >= 3, info
    assert planned._pai_opt_steps == eager._pai_opt_steps == 2 * steps
    for oe, og in zip(eager._all_optimizers(), planned._all_optimizers()):
        assert og.total_steps == oe.total_steps == steps
        se, sg = oe.state_dict()["state"], og.state_dict()["state"]
        assert int(next(iter(sg.values()))["step"]) == int(next(iter(se.values()))["step"]) == steps


def test_plug_in_network_steps_aside(pai):
    """A ``unet`` the library does not know (any nn.Module, reference README.md:23) launches torch's own kernels inside the
    step: the recorder sees them, refuses, and every call runs the eager step -- never a partial replay."""
    from thesis_pai_reconstruction_amd.plan import PlannedStep

    class Plug(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(1, 1, 3, padding=1)

        def forward(self, x):
            return torch.tanh(self.c(x))

    torch.manual_seed(0)
    m = pai.UnetWrapper(Plug(), "gan")
    m.to(DEV)
    m.train()
    ps = PlannedStep(m, warmup=3)
    for s in range(6):
        ps(tuple(t.to(DEV) for t in synth_batch(300 + s, 2, 64)), s)
    torch.cuda.synchronize()
    assert ps.disabled is not None and ps.replays == 0 and not ps.plans, ps.describe()
    assert m._pai_opt_steps == 12 and np.isfinite(float(m.logged["loss"]))


def test_plan_abi_streams_events_and_adam(pai):
    """The C ABI by itself: launches on two streams with a pai_stream_wait edge and a caller-owned event are recorded
    while they run and replayed on new data; a recorded pai_adam launch replays with the bias correction of
    step0 + step_delta, bit-identical to the eager launch of that step."""
    from thesis_pai_reconstruction_amd import ops
    dev = torch.device(DEV)
    n = 1 << 16
    a = torch.randn(n, device=dev)
    b = torch.randn(n, device=dev)
    out1 = torch.empty(n, device=dev)
    out2 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    ev = ops.Event()
    plan = ops.Plan()
    with plan.recording():
        ops.add_act(torch.float32, a, b, ops.ACT_RELU, out1)           # main
        ops.stream_wait(side, main)
        with torch.cuda.stream(side):
            ops.cast(out1, out2)                                       # side, behind the add
            ev.record(side)
        ev.wait(main)
        ops.zero_multi([out1])                                         # main, behind the cast (it overwrites its input)
    torch.cuda.synchronize()
    info = plan.info()
    assert info == {"launches": 3, "waits": 2, "streams": 2, "runs": 0}, info
    assert torch.equal(out2.float(), torch.relu(a + b).bfloat16().float()) and float(out1.abs().max()) == 0.0
    a.copy_(torch.randn(n, device=dev))
    want = torch.relu(a + b).bfloat16().float()
    plan.run()
    torch.cuda.synchronize()
    assert torch.equal(out2.float(), want) and float(out1.abs().max()) == 0.0
    assert plan.info()["runs"] == 1

    # Adam: record step 3, replay as steps 4 and 7
    p0 = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev)
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-7

    def fresh():
        return p0.clone(), torch.full((n,), 0.01, device=dev), torch.full((n,), 0.02, device=dev)
    p, m, v = fresh()
    plan2 = ops.Plan()
    with plan2.recording():
        ops.adam(p, g, m, v, lr, b1, b2, eps, 3)
    for delta in (1, 4):
        pe, me, ve = fresh()
        ops.adam(pe, g, me, ve, lr, b1, b2, eps, 3 + delta)
        q, mq, vq = fresh()
        p.copy_(q), m.copy_(mq), v.copy_(vq)
        plan2.run(delta)
        torch.cuda.synchronize()
        assert torch.equal(p, pe) and torch.equal(m, me) and torch.equal(v, ve), delta
    with pytest.raises(ops.PaiError):
        plan2.run(-1)


def test_launch_timer_rides_on_one_launch(pai):
    """pai_profile_arm: the armed events time exactly the next launch (its own start / stop events), are consumed by it, and
    an unused arming can be withdrawn."""
    from thesis_pai_reconstruction_amd import ops

    t = torch.ones(1 << 24, device=DEV)
    ops.scale_(t, 1.0)
    torch.cuda.synchronize()
    timer = ops.LaunchTimer()
    timer.arm()
    ops.scale_(t, 2.0)          # timed
    ops.scale_(t, 2.0)          # not timed: the arming was consumed
    torch.cuda.synchronize()
    ms = timer.elapsed_time()
    # 128 MB of traffic: 16-40 us on this part; two launches, or a bracket with gaps, would read far longer
    assert 0.005 < ms < 0.2, ms
    assert float(t[0]) == 4.0
    timer.arm()
    ops.LaunchTimer.disarm()
    ops.scale_(t, 0.5)
    torch.cuda.synchronize()
    assert abs(timer.elapsed_time() - ms) < 1e-9      # the withdrawn arming did not touch the events
