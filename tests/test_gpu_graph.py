"""hipGraph replay of the whole GAN step (graph.GraphedStep) against the eager step: same losses / metrics step by
step, same parameters afterwards, the Adam step count and BatchNorm bookkeeping advance on the device, and replay costs
one launch of host time (reference step: models/wrapper.py:117-162)."""
import time

import numpy as np
import pytest
import torch

import oracle
from oracle.gen_golden import synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(pai, mults, seed, dtype):
    m = pai.Pix2Pix(1, 1, tuple(mults), 0.0, "gan")
    m.unet.load_state_dict(oracle.init_state_portable(oracle.make_unet_state(1, 1, tuple(mults)), seed, perturb_bn=True))
    m.discriminator.load_state_dict(oracle.init_state_portable(oracle.make_disc_state(1), seed + 1))
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m


def _sync_training_state(src, dst):
    """dst <- src, in place (a captured graph holds the addresses): parameters, BatchNorm buffers, Adam moments."""
    with torch.no_grad():
        sd = dst.state_dict()
        for k, v in src.state_dict().items():
            sd[k].copy_(v)
        for os_, od in zip(src._all_optimizers(), dst._all_optimizers()):
            a_s, a_d = os_._engine.arena(), od._engine.arena()
            if getattr(a_s, "mflat", None) is not None and getattr(a_d, "mflat", None) is not None:
                a_d.mflat.copy_(a_s.mflat)
                a_d.vflat.copy_(a_s.vflat)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_graphed_step_matches_eager(pai, dtype):
    """Every step starts from the SAME state in both models (the eager model's, copied in place after each comparison):
    a GAN step amplifies last-bit noise (fp32 atomics of the bias / thin-layer gradients order differently from run to
    run) by 2-3x per step -- two eager runs of seven free-running steps differ by up to 8e-2 in d_loss -- so free-running
    models can only be held to a bound that says nothing.  Step by step the bound is tight."""
    from thesis_pai_reconstruction_amd.graph import GraphedStep
    mults, n, size, steps = (1, 2, 4, 8), 4, 64, 7
    batches = [tuple(t.to(DEV) for t in synth_batch(100 + s, n, size)) for s in range(steps)]
    eager, graphed = (_build(pai, mults, 3, dtype) for _ in range(2))
    gs = GraphedStep(graphed, warmup=2)
    for s, b in enumerate(batches):
        eager.logged, graphed.logged = {}, {}
        eager.training_step(b, s)
        gs(b, s)
        torch.cuda.synchronize()
        assert gs.disabled is None, gs.disabled
        for k, v in eager.logged.items():
            a, g = float(v), float(graphed.logged[k])
            assert abs(a - g) <= 1e-4 * max(1.0, abs(a)), (s, k, a, g)
        for (k, p), (_, q) in zip(eager.state_dict().items(), graphed.state_dict().items()):
            if k.endswith("num_batches_tracked"):
                assert int(p) == int(q) == 2 * (s + 1), k
            else:
                # one step apart from identical states: +-lr per element at most where noise flips a sign Adam's first
                # steps normalise (lr = 2e-4), far less in the mean
                d = (p.float() - q.float()).abs()
                assert float(d.max()) <= 4.1e-4 and float(d.mean()) <= 1e-5, (s, k, float(d.max()), float(d.mean()))
        _sync_training_state(eager, graphed)
    assert gs.graph is not None and gs.opt_steps_per_replay == 2
    assert graphed._pai_opt_steps == eager._pai_opt_steps == 2 * steps
    for oe, og in zip(eager._all_optimizers(), graphed._all_optimizers()):
        assert og.total_steps == oe.total_steps == steps and int(og._dev_step) == steps
        se, sg = oe.state_dict()["state"], og.state_dict()["state"]
        assert int(next(iter(sg.values()))["step"]) == int(next(iter(se.values()))["step"]) == steps


def test_graph_replay_is_one_launch_of_host_time(pai):
    from thesis_pai_reconstruction_amd.graph import GraphedStep
    m = _build(pai, (1, 2, 4, 8, 8), 5, torch.bfloat16)
    b = tuple(t.to(DEV) for t in synth_batch(7, 8, 128))
    gs = GraphedStep(m, warmup=2)
    for s in range(4):
        gs(b, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10):
        gs(b, s)
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10):
        m.training_step(b, s)
    eager = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print(f"host issue per step: graph {host * 1e3:.3f} ms, eager {eager * 1e3:.3f} ms")
    assert host < 1e-3 and host < 0.5 * eager
    assert np.isfinite(float(m.logged["loss"]))


@pytest.mark.parametrize("family", ["resnext_unet", "trans_unet"])
def test_composable_families_capture_and_replay(pai, family):
    """The residual / Trans U-Nets (op-level path of nnops.py, MultiAdam) as one hipGraph: warm-up and capture on one side
    stream (autograd's AccumulateGrad nodes), the Adam step count on the device (pai_adam_multi_dev).  Graph against an
    eager twin from the same weights: same losses step by step, same step counts (reference models/res_unet.py:238-335,
    models/trans_unet.py:35-117)."""
    import copy
    from thesis_pai_reconstruction_amd.graph import GraphedStep
    torch.manual_seed(0)
    if family == "resnext_unet":
        eager = pai.ResUnetGAN(1, 1, "next", (1, 2), 0.0, "gan")
    else:
        eager = pai.TransUnetGAN(1, 1, (1, 2), 2, 0.0, "gan")
    eager.to(DEV)
    eager.set_precision("bf16-mixed")
    eager.train()
    graphed = copy.deepcopy(eager)
    gs = GraphedStep(graphed, warmup=2)
    size = 64 if family == "resnext_unet" else 256        # the TransUNet is built for 256 x 256 inputs
    batches = [tuple(t.to(DEV) for t in synth_batch(300 + s, 2, size)) for s in range(6)]
    for s, b in enumerate(batches):
        eager.logged, graphed.logged = {}, {}
        eager.training_step(b, s)
        gs(b, s)
        torch.cuda.synchronize()
        assert gs.disabled is None, gs.disabled
        for k, v in eager.logged.items():
            a, g = float(v), float(graphed.logged[k])
            assert abs(a - g) <= 5e-2 * max(1.0, abs(a)), (s, k, a, g)
    assert gs.graph is not None and gs.opt_steps_per_replay == 2
    for oe, og in zip(eager._all_optimizers(), graphed._all_optimizers()):
        se, sg = oe.state_dict()["state"], og.state_dict()["state"]
        assert int(next(iter(sg.values()))["step"]) == int(next(iter(se.values()))["step"]) == len(batches)
        assert int(og._dev_step) == len(batches)
