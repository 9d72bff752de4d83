"""GPU parity of the residual U-Net family (reference models/res_unet.py) through the plugin
surface (ResUnetGAN, res_type "next" / "18" / "50") against fixtures recorded from the REAL reference and the CPU
oracle run live.  fp32 mode within 1e-4 relative; bf16 mode a looser bound."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle.fingerprint import fingerprint, fingerprint_close
from oracle.gen_golden import synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def build(pai, res_type, mults, loss_type, seed, dtype=torch.float32, dropout=0.0):
    m = pai.ResUnetGAN(in_channels=1, out_channels=1, res_type=res_type, channel_mults=tuple(mults), dropout=dropout,
                       loss_type=loss_type)
    g = oracle.init_state_portable(oracle.make_res_unet_state(1, 1, res_type, tuple(mults)), seed, perturb_bn=True)
    m.unet.load_state_dict(g, strict=True)
    d = None
    if loss_type == "gan":
        d = oracle.init_state_portable(oracle.make_disc_state(1), seed + 1)
        m.discriminator.load_state_dict(d, strict=True)
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m, g, d


def _fp_err(got, want_fp, rtol):
    ok, worst = fingerprint_close(fingerprint(got), want_fp, rtol)
    return ok, worst


@pytest.mark.parametrize("name", ["ref_resnext_forward_tiny", "ref_res18_forward_tiny", "ref_res50_forward_tiny",
                                  "ref_resv2_forward_tiny", "ref_resnext_forward_mid",
                                  # BASELINE configs[3] at its real width: res_type "next", channel_mults (1,2,4,8,8,8,8,8)
                                  "ref_resnext_forward_full"])
def test_forward_matches_reference_fixture(pai, golden_dir, name):
    z = _load(golden_dir, name)
    seed, n, size, fam = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), str(z["meta.family"])
    m, _, _ = build(pai, fam[3:], [int(v) for v in z["meta.mults"]], "gan", seed)
    x, t = synth_batch(seed + 100, n, size)
    x = x.to(DEV)
    with torch.no_grad():
        pred = m.unet(x)
        lf = m.discriminator(x, pred)
    want = torch.from_numpy(z["pred_full"])
    # 3-level fixtures at 1e-4; the 5-level one (10 residual blocks, BatchNorm over 32 samples at the bottom) at 5e-4; the
    # 8-level one at 512 x 512 (16 residual blocks, 32 samples at the bottom) at 5e-4 + 3 x the distance of the reference's
    # OWN fp32 prediction from its fp64 one (oracle/gen_f64_floor.py: 1.1e-3 -- fp32 rounding through 16 BatchNorms)
    ftol = 1e-4 if name.endswith("tiny") else 5e-4
    if os.path.exists(os.path.join(golden_dir, name + "_f64floor.npz")):
        ftol += 3.0 * float(_load(golden_dir, name + "_f64floor")["floor.pred"])
    assert float((pred.cpu() - want).abs().max()) < ftol * float(want.abs().max())
    w = torch.from_numpy(z["logits_fake_full"])
    # (the 32 x 32 fixtures give one logit of ~1e-3 per sample: bound relative to the activations that form it)
    assert float((lf.cpu() - w).norm()) < max(ftol / 5, 1e-4) * max(float(w.norm()), 1e-2)     # (logits of the noisy prediction)
    m.eval()                                   # eval mode (running statistics) against the live oracle
    g = {k: v.detach().cpu().clone() for k, v in m.unet.state_dict().items()}
    with torch.no_grad():
        pe = m.unet(x)
        we = oracle.res_unet_forward(g, x.cpu(), training=False)
    assert float((pe.cpu() - we).abs().max()) < ftol * float(we.abs().max())


def _check_step(m, z, s, gtol, floor=None):
    for k, v in m.logged.items():
        want = float(z[f"step{s}.log.{k}"])
        # after the first update the kink flips described below have moved the parameters apart by ~lr: 1e-3 from there
        assert abs(float(v) - want) <= (1e-4 if s == 0 else 1e-3) * max(1.0, abs(want)), (s, k, float(v), want)
    names = [k for k, _ in m.unet.named_parameters()]
    gmax = max(float(z[f"step{s}.ggrad.{k}"][3]) for k in names)
    bad = []
    for k, p in m.unet.named_parameters():
        want = z[f"step{s}.ggrad.{k}"]
        if p.dim() == 1 and float(want[3]) < 1e-4 * gmax:
            # analytically zero gradients (conv bias in front of a BatchNorm; per-channel shifts whose every consumer is a
            # 1x1 conv -> BatchNorm pair): cancellation noise in the reference, see tests/test_oracle_golden.py
            assert float(p.grad.abs().max()) < 1e-3 * gmax, (s, k)
            continue
        extra = 3.0 * float(floor["floor." + k]) if floor is not None and s == 0 else 0.0
        ok, worst = _fp_err(p.grad, want, (gtol if p.numel() > 1 else 2e-2) + extra)
        if not ok:
            bad.append((k, worst))
    assert not bad, (s, bad[:6])


@pytest.mark.parametrize("name", ["ref_resnext_gan_tiny", "ref_res18_gan_tiny", "ref_res50_gan_tiny", "ref_resv2_gan_tiny",
                                  "ref_resnext_gan_full"])
def test_gan_training_step_matches_reference_fixture(pai, golden_dir, name):
    z = _load(golden_dir, name)
    seed, n, size, steps = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), int(z["meta.steps"])
    fam = str(z["meta.family"])
    m, g, d = build(pai, fam[3:], [int(v) for v in z["meta.mults"]], "gan", seed)
    x, t = synth_batch(seed + 100, n, size)
    batch = (x.to(DEV), t.to(DEV))
    for s in range(steps):
        m.logged = {}
        m.training_step(batch, s)
        torch.cuda.synchronize()
        # Step 0 is the parity bar (tests/test_gpu_model.py: later steps start from sign-SGD-perturbed parameters).
        # Against the live oracle every gradient is within 1e-5 (L2) UNLESS a ReLU flips: the 32 x 32 x 64..128-channel
        # activations of these nets have 260-520 K elements each -- beyond what the fixture's seed search keeps away from
        # the kink -- and at our forward noise of ~1e-5 (fp32 sums of up to 1152 terms in another order) a few
        # pre-activations per tensor change sign; one flip moves every upstream gradient by ~1/sqrt(numel) ~ 3e-3
        # (scripts/debug_res_grad.py shows the step where it enters).  Hence 6e-3, plus the reference's own fp32
        # distance from fp64 (oracle/gen_f64_floor.py; 2-5e-3 for res_type "50", whose 16-channel bottlenecks are tiny).
        floor = _load(golden_dir, name + "_f64floor") if os.path.exists(os.path.join(golden_dir, name + "_f64floor.npz")) else None
        _check_step(m, z, s, 6e-3 if s == 0 else 0.3 * s, floor)
    for k, v in m.unet.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 2 * steps       # two BatchNorm updates per GAN step
    m.eval()
    m.logged = {}
    with torch.no_grad():
        m.validation_step(batch, 0)
    for k, v in m.logged.items():
        want = float(z[f"val.log.{k}"])
        assert abs(float(v) - want) <= 2e-3 * max(1.0, abs(want)), (k, float(v), want)


def test_dropout2d_step_matches_reference_fixture(pai, golden_dir):
    """Dropout2d(0.5) behind the widest decoder blocks (reference models/res_unet.py:230,286-289), masks replayed from
    the oracle's redraw of the reference's generator state."""
    z = _load(golden_dir, "ref_resnext_gan_dropout_tiny")
    seed, n, size, steps = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), int(z["meta.steps"])
    mults, p = [int(v) for v in z["meta.mults"]], float(z["meta.dropout"])
    m, g, d = build(pai, "next", mults, "gan", seed, dropout=p)
    assert not m.unet.supports_forward_reuse
    x, t = synth_batch(seed + 100, n, size)
    batch = (x.to(DEV), t.to(DEV))
    og, od = oracle.AdamState(), oracle.AdamState()
    for s in range(steps):
        torch.manual_seed(1000 + s)
        mask_log = []
        oracle.gan_training_step(g, d, og, od, x, t, dropout=p, mask_log=mask_log)
        queue = list(mask_log)
        assert queue

        def replay(j, N, C, rate, device):
            jj, mk = queue.pop(0)
            assert jj == j and rate == p and mk.shape[:2] == (N, C)
            return mk.reshape(N, C).to(device)

        m.unet.dropout_mask_fn = replay
        m.logged = {}
        m.training_step(batch, s)
        torch.cuda.synchronize()
        assert not queue
        _check_step(m, z, s, 5e-3 if s == 0 else 0.3 * s)


def test_bf16_mode_tracks_fp32(pai, golden_dir):
    """bf16 storage: the MFMA kernels behind the 3x3 / 1x1 / block-diagonal grouped convolutions.  At random
    initialisation in_conv's 64 channels are a bias plus a small signal of rank <= 9, and every block starts with a
    1x1 conv -> BatchNorm that removes the constant again: the 2^-9 storage rounding of the total becomes ~1 % of the
    signal per block (scripts/debug_res.py lists it level by level; fp32 mode stays at 7e-5 through 10 blocks).  So
    the bound here is loose on the output and the check that matters is that training makes progress."""
    z = _load(golden_dir, "ref_resnext_forward_tiny")
    seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
    mults = [int(v) for v in z["meta.mults"]]
    m, _, _ = build(pai, "next", mults, "gan", seed, dtype=torch.bfloat16)
    x, t = synth_batch(seed + 100, n, size)
    batch = (x.to(DEV), t.to(DEV))
    with torch.no_grad():
        p16 = m.unet(batch[0])
    want = torch.from_numpy(z["pred_full"])
    # noise-dominated at random initialisation (docstring): a sanity bound only; the bounds that mean something are
    # the trajectory ones below (and tests/test_gpu_configs.py at the full configs[3] size)
    assert float((p16.cpu() - want).norm() / want.norm()) < 0.25
    m32, _, _ = build(pai, "next", mults, "gan", seed)
    first = None
    for s in range(4):
        logs = []
        for mm in (m, m32):
            mm.logged = {}
            mm.training_step(batch, s)
            logs.append({k: float(v) for k, v in mm.logged.items()})
        vals, vals32 = logs
        assert all(np.isfinite(v) for v in vals.values()), vals
        for k in ("loss", "d_loss", "train_rmse", "train_psnr"):
            # bf16 storage follows the fp32 parity path step by step (measured <= 0.4 % at the configs[3] size; on this tiny,
            # noise-dominated network up to ~2 % by the fourth step, differently from run to run: the fp32 atomics of the
            # thin layers' gradients order the roundings)
            assert abs(vals[k] - vals32[k]) <= 0.04 * max(abs(vals32[k]), 1.0), (s, k, vals[k], vals32[k])
        first = first or vals
    assert vals["loss"] < first["loss"] and vals["train_rmse"] < first["train_rmse"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_side_stream_weight_gradients_equal_main_stream_ones(pai, golden_dir, dtype):
    """nnops._WgradStream: the weight gradients run on a second stream; a bare ``loss.backward()`` (no manual_backward, no
    optimizer) must hand back the gradients of the single-stream order -- the join is the autograd engine's end-of-pass
    callback.  A layer whose parameters already hold a gradient stays on the main stream (autograd accumulates there)."""
    from thesis_pai_reconstruction_amd import nnops

    z = _load(golden_dir, "ref_resnext_forward_tiny")
    seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
    mults = [int(v) for v in z["meta.mults"]]
    x, t = synth_batch(seed + 7, n, size)
    x, t = x.to(DEV), t.to(DEV)
    grads = {}
    assert nnops.WGRAD.on
    for mode in ("side", "main", "accumulate"):
        m, _, _ = build(pai, "next", mults, "mse", seed, dtype=dtype)
        nnops.WGRAD.on = mode != "main"
        try:
            loss = m.loss(x, m.unet(x), t)
            loss.backward()
            if mode == "accumulate":          # second pass onto existing gradients
                m.loss(x, m.unet(x), t).backward()
            assert not nnops.WGRAD.pending and not nnops.WGRAD.keep
        finally:
            nnops.WGRAD.on = True
        grads[mode] = {k: p.grad.detach().clone() for k, p in m.unet.named_parameters() if p.grad is not None}
    assert grads["side"].keys() == grads["main"].keys() and len(grads["side"]) > 20
    scale = max(float(g.norm()) for g in grads["main"].values())
    for k, g in grads["main"].items():
        # a few kernels sum partial results with fp32 atomics (thin in_conv / out layers): equal up to summation order
        # (and a bias in front of a BatchNorm has a gradient that is rounding noise around zero: absolute floor)
        err = float((grads["side"][k] - g).norm())
        assert err <= 2e-6 * float(g.norm()) + 1e-7 * scale, (k, err, float(g.norm()), scale)
        # BatchNorm buffers moved between the two passes of the accumulate run, so only the scale is compared
        ratio = float(grads["accumulate"][k].norm() / g.norm().clamp_min(1e-30))
        assert g.norm() == 0 or 0.5 < ratio < 4.0, (k, ratio)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_parameter_used_by_two_nodes_of_one_graph(pai, dtype):
    """a convolution applied TWICE in one graph.  The first use's weight gradient goes to the side stream; the
    second use runs on the main stream (autograd adds the two there, before the end-of-pass join) and must first wait for
    the side stream -- otherwise the sum reads a gradient that is still being written.  Against PAI_NO_OVERLAP-style
    single-stream issue, over several repetitions (a race shows up as run-to-run differences)."""
    from thesis_pai_reconstruction_amd import nnops

    torch.manual_seed(11)
    conv = torch.nn.Conv2d(64, 64, 1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(64).to(DEV)
    big = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)      # a long weight gradient in front of the re-used layer
    x = torch.randn(8, 64, 128, 128, device=DEV)
    got = {}
    assert nnops.WGRAD.on
    for mode in ("main", "side", "side", "side"):
        for m in (conv, bn, big):
            for p in m.parameters():
                p.grad = None
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        nnops.WGRAD.on = mode != "main"
        try:
            h = nnops.to_nhwc(x, dtype)
            h = nnops.conv_bn_act(h, big, None, nnops.ACT_NONE, True, 1, dtype)
            h = nnops.conv_bn_act(h, conv, bn, nnops.ACT_RELU, True, 1, dtype)     # first use
            h = nnops.conv_bn_act(h, conv, bn, nnops.ACT_RELU, True, 1, dtype)     # second use of the same parameters
            h.float().square().mean().backward()
            assert not nnops.WGRAD.pending
        finally:
            nnops.WGRAD.on = True
        torch.cuda.synchronize()
        got.setdefault(mode, []).append({"conv": conv.weight.grad.clone(), "big": big.weight.grad.clone(),
                                         "gamma": bn.weight.grad.clone()})
    ref = got["main"][0]
    for run in got["side"]:
        for k, g in ref.items():
            err = float((run[k] - g).norm())
            assert err <= 1e-5 * float(g.norm()) + 1e-12, (k, err, float(g.norm()))


def test_batchnorm_on_load_equals_batchnorm_as_a_pass(pai, monkeypatch, res_type="next"):
    """bf16: where the next convolution of a block can read its input through a prologue (``nnops.can_prologue``: the grouped
    3 x 3 and the 1 x 1 behind it of a ResNeXt block at >= 16384 pixels; reference models/res_unet.py:143-147), the BatchNorm +
    ReLU in between never writes its tensor.  Against PAI_NO_PROLOGUE=1 (every BatchNorm a pass of its own): the prediction bit
    for bit (the prologue forms the same bf16 values), the gradients up to the atomics of the weight-gradient splits."""
    from thesis_pai_reconstruction_amd import nnops, ops

    mults, n, size, seed = (1, 2), 2, 128, 5
    x, t = synth_batch(seed + 3, n, size)
    x, t = x.to(DEV), t.to(DEV)
    out, grads, calls = {}, {}, {}
    real = ops.conv_fwd_pro
    for mode in ("pass", "load"):
        monkeypatch.setenv("PAI_NO_PROLOGUE", "1" if mode == "pass" else "0")
        count = [0]

        def counted(*a, **k):
            count[0] += 1
            return real(*a, **k)
        monkeypatch.setattr(ops, "conv_fwd_pro", counted)
        m, _, _ = build(pai, res_type, mults, "mse", seed, dtype=torch.bfloat16)
        pred = m.unet(x)
        m.loss(x, pred, t).backward()
        nnops.join_wgrads()
        torch.cuda.synchronize()
        out[mode], calls[mode] = pred.detach().clone(), count[0]
        grads[mode] = {k: p.grad.detach().clone() for k, p in m.unet.named_parameters() if p.grad is not None}
        m.eval()                         # eval mode: the prologue carries the running statistics' scale / shift
        with torch.no_grad():
            out[mode + "_eval"] = m.unet(x).detach().clone()
    assert torch.equal(out["load_eval"], out["pass_eval"])
    assert calls["pass"] == 0
    assert calls["load"] >= 2, calls       # the level-0 encoder block (2 x 128 x 128 pixels): grouped 3 x 3 and the last 1 x 1
    assert torch.equal(out["load"], out["pass"])
    assert grads["load"].keys() == grads["pass"].keys()
    scale = max(float(g.norm()) for g in grads["pass"].values())
    for k, g in grads["pass"].items():
        err = float((grads["load"][k] - g).norm())
        assert err <= 1e-5 * float(g.norm()) + 1e-6 * scale, (k, err, float(g.norm()), scale)


def test_one_pack_launch_per_forward_equals_a_pack_per_layer(pai, monkeypatch):
    """``nnops.prepack``: the bf16 packs of all dense pointwise convolutions from ONE pai_pack_weights_multi launch at the start
    of the forward pass -- same packs, so the same prediction and gradients as with a pack launch in front of each layer, and
    the table is empty again when the pass is over (nothing stale can be picked up later)."""
    from thesis_pai_reconstruction_amd import nnops

    mults, n, size, seed = (1, 2, 4), 2, 64, 9
    x, t = synth_batch(seed, n, size)
    x, t = x.to(DEV), t.to(DEV)
    res = {}
    for mode in ("multi", "single"):
        monkeypatch.setenv("PAI_NO_PREPACK", "0" if mode == "multi" else "1")
        m, _, _ = build(pai, "next", mults, "mse", seed, dtype=torch.bfloat16)
        pred = m.unet(x)
        assert not nnops._PREPACK
        m.loss(x, pred, t).backward()
        nnops.join_wgrads()
        torch.cuda.synchronize()
        res[mode] = (pred.detach().clone(), {k: p.grad.detach().clone() for k, p in m.unet.named_parameters()})
    assert torch.equal(res["multi"][0], res["single"][0])
    scale = max(float(g.norm()) for g in res["single"][1].values())
    for k, g in res["single"][1].items():
        assert float((res["multi"][1][k] - g).norm()) <= 2e-6 * float(g.norm()) + 1e-7 * scale, k
