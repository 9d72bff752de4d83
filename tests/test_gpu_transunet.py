"""GPU parity of the TransUNet path (reference models/trans_unet.py): the token ops of csrc/vit.hip
against plain PyTorch-CPU fp32 ops, and the TransUnetGAN plugin class against fixtures recorded from the REAL reference
(oracle/gen_golden.py --trans) and the live oracle.  fp32 mode within 1e-4 relative on outputs; bf16 a looser bound."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from _gpu_util import dev, q, rel_err, rnd
from oracle.fingerprint import fingerprint, fingerprint_close
from oracle.gen_golden import synth_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])


def _tol(dtype, f32=1e-5, bf16=1.2e-2):
    return f32 if dtype == torch.float32 else bf16


def _d(t, dtype):
    return t.to(DEV).to(dtype).contiguous()


@DTYPES
@pytest.mark.parametrize("M,D,P", [(12, 96, 4), (64, 512, 16), (7, 1000, 1), (128, 4096, 4), (8, 8192, 4)])
@pytest.mark.parametrize("fused", [False, True], ids=["plain", "res+post"])
def test_layernorm(pai, dtype, M, D, P, fused):
    from thesis_pai_reconstruction_amd import nnops
    x, r = q(rnd((M, D), 1) * 1.5 + 0.2, dtype), q(rnd((M, D), 2), dtype)
    gamma, beta, post = 1 + 0.1 * rnd((D,), 3), 0.1 * rnd((D,), 4), rnd((1, P, D), 5)
    gy = q(rnd((M, D), 6), dtype)
    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    gr, br, pr = (t.clone().requires_grad_(True) for t in (gamma, beta, post))
    s = xr
    if fused:       # the kernel normalises the STORED (rounded) sum; the rounding passes gradients straight through
        s = xr + rr
        s = s + (q(s.detach(), dtype) - s.detach())
    want = F.layer_norm(s, (D,), gr, br, 1e-5)
    if fused:
        want = (want.view(M // P, P, D) + pr).view(M, D)
    (want * gy).sum().backward()

    xd, rd = _d(x, dtype).requires_grad_(True), _d(r, dtype).requires_grad_(True)
    gd, bd, pd = (t.to(DEV).requires_grad_(True) for t in (gamma, beta, post))
    got = nnops.LayerNorm.apply(xd, rd if fused else None, gd, bd, 1e-5, pd if fused else None)
    (got.float() * gy.to(DEV)).sum().backward()
    tol = _tol(dtype)
    assert rel_err(got.float().cpu(), want.detach()) < tol
    assert rel_err(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_err(gd.grad.cpu(), gr.grad) < tol and rel_err(bd.grad.cpu(), br.grad) < tol
    if fused:
        assert rel_err(rd.grad.float().cpu(), rr.grad) < tol
        assert rel_err(pd.grad.cpu(), pr.grad) < tol


@DTYPES
def test_gelu(pai, dtype):
    from thesis_pai_reconstruction_amd import nnops
    z = q(rnd((37, 256), 7) * 2.0, dtype)
    gy = q(rnd((37, 256), 8), dtype)
    zr = z.clone().requires_grad_(True)
    want = F.gelu(zr)
    (want * gy).sum().backward()
    zd = _d(z, dtype).requires_grad_(True)
    got = nnops.GELU.apply(zd)
    (got.float() * gy.to(DEV)).sum().backward()
    assert rel_err(got.float().cpu(), want.detach()) < _tol(dtype, 1e-6, 4e-3)
    assert rel_err(zd.grad.float().cpu(), zr.grad) < _tol(dtype, 1e-6, 6e-3)


@DTYPES
@pytest.mark.parametrize("S,B,heads,hd", [(4, 16, 8, 64), (3, 4, 8, 128), (32, 4, 8, 512), (5, 2, 2, 24), (70, 1, 1, 16)])
def test_mha_core_matches_torch_mha(pai, dtype, S, B, heads, hd):
    """Against F.multi_head_attention_forward itself (identity projections folded out: qkv are given)."""
    from thesis_pai_reconstruction_amd import nnops
    E = heads * hd
    qkv = q(rnd((S * B, 3 * E), 9) * 0.7, dtype)
    gy = q(rnd((S * B, E), 10), dtype)
    qr = qkv.clone().requires_grad_(True)
    qq, kk, vv = (c.reshape(S, B * heads, hd).transpose(0, 1) for c in qr.view(S, B, 3 * E).chunk(3, dim=-1))
    want = F.scaled_dot_product_attention(qq, kk, vv).transpose(0, 1).reshape(S * B, E)
    (want * gy).sum().backward()
    qd = _d(qkv, dtype).requires_grad_(True)
    got = nnops.MHACore.apply(qd, S, B, heads)
    (got.float() * gy.to(DEV)).sum().backward()
    assert rel_err(got.float().cpu(), want.detach()) < _tol(dtype, 2e-6, 6e-3)
    assert rel_err(qd.grad.float().cpu(), qr.grad) < _tol(dtype, 4e-6, 1.2e-2)


@pytest.mark.parametrize("S,B,heads,hd,drop", [(32, 4, 8, 512, False), (32, 16, 8, 128, True), (7, 3, 2, 64, True), (1, 2, 4, 32, False)])
def test_mha_matrix_core_path_against_the_vector_path(pai, S, B, heads, hd, drop):
    """bf16 storage, S <= 32, head dim % 32 == 0: pai_mha_fwd / pai_mha_bwd run mha_fwd_mfma_k / mha_bwd_mfma_k
    (v_mfma_f32_32x32x16_bf16 for K Q^T, P V, dO V^T, dS K, dS^T Q, P^T dO; BASELINE configs[4] "attention-MFMA",
    reference models/trans_unet.py:151-161 through nn.MultiheadAttention).  With the tunable mha_mfma = 0 the same calls take
    the vector-ALU kernels: outputs, kept probabilities and all three gradients of the two paths agree to bf16 rounding,
    with and without an attention-dropout mask, full and ragged sequence lengths."""
    from thesis_pai_reconstruction_amd import ops
    E = heads * hd
    dt = torch.bfloat16
    qkv = _d(q(rnd((S * B, 3 * E), 21) * 0.7, dt), dt)
    gy = _d(q(rnd((S * B, E), 22), dt), dt)
    mask = None
    if drop:
        torch.manual_seed(5)
        mask = (torch.bernoulli(torch.full((B * heads, S, S), 0.7)) / 0.7).to(DEV).contiguous()
    res = []
    for mfma in (1, 0):
        ops.set_tunable("mha_mfma", mfma)
        try:
            out = torch.empty(S * B, E, dtype=dt, device=DEV)
            probs = torch.full((B * heads * S * S,), float("nan"), dtype=torch.float32, device=DEV)
            ops.mha_fwd(dt, qkv, S, B, heads, hd, out, probs, mask)
            dqkv = torch.full_like(qkv, float("nan"))
            ds = torch.empty_like(probs)
            ops.mha_bwd(dt, gy, qkv, probs, S, B, heads, hd, dqkv, ds, mask)
            torch.cuda.synchronize()
        finally:
            ops.set_tunable("mha_mfma")
        assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(probs).all()) and bool(torch.isfinite(dqkv.float()).all())
        res.append((out.float().cpu(), probs.cpu(), dqkv.float().cpu()))
    (o1, p1, g1), (o0, p0, g0) = res
    assert rel_err(p1, p0) < 2e-3 and float((p1.view(-1, S).sum(1) - 1).abs().max()) < 1e-5      # fp32 softmax of bf16 products
    assert rel_err(o1, o0) < 8e-3
    for part, name in ((slice(0, E), "dQ"), (slice(E, 2 * E), "dK"), (slice(2 * E, 3 * E), "dV")):
        assert rel_err(g1[:, part], g0[:, part]) < 1.5e-2, name


@DTYPES
@pytest.mark.parametrize("M,K,O", [(16, 96, 40), (64, 512, 1536), (128, 1024, 2048), (12, 256, 256), (128, 4096, 2048), (100, 132, 72)])
def test_linear(pai, dtype, M, K, O):
    from thesis_pai_reconstruction_amd import nnops
    x, w, b = q(rnd((M, K), 11), dtype), q(rnd((O, K), 12) * 0.05, dtype), rnd((O,), 13) * 0.1
    gy = q(rnd((M, O), 14), dtype)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    want = F.linear(xr, wr, br)
    (want * gy).sum().backward()
    xd = _d(x, dtype).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    got = nnops.Linear.apply(xd, wd, bd)
    (got.float() * gy.to(DEV)).sum().backward()
    tol = _tol(dtype, 2e-5, 8e-3)
    assert rel_err(got.float().cpu(), want.detach()) < tol
    assert rel_err(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_err(wd.grad.cpu(), wr.grad) < tol and rel_err(bd.grad.cpu(), br.grad) < tol


@DTYPES
def test_strided_convs_through_subsample(pai, dtype):
    """Conv2d(k3, s2, p1) and Conv2d(k1, s2) composed from the stride-1 kernels and Subsample2, and BNAct on the
    subsampled tensor, against F.conv2d(stride=2) -> F.batch_norm -> relu autograd."""
    from thesis_pai_reconstruction_amd import nnops
    from thesis_pai_reconstruction_amd.ops import ACT_NONE, ACT_RELU
    N, C, H, W, Co = 3, 16, 12, 20, 24
    x = q(rnd((N, C, H, W), 15), dtype)
    w3, w1 = q(rnd((C, C, 3, 3), 16) * 0.1, dtype), q(rnd((Co, C, 1, 1), 17) * 0.2, dtype)
    gamma, beta = 1 + 0.1 * rnd((C,), 18), 0.1 * rnd((C,), 19)
    gy3, gy1 = q(rnd((N, C, H // 2, W // 2), 20), dtype), q(rnd((N, Co, H // 2, W // 2), 21), dtype)
    xr, w3r, w1r, gr, br = (t.clone().requires_grad_(True) for t in (x, w3, w1, gamma, beta))
    rm, rv = torch.zeros(C), torch.ones(C)
    y3 = F.relu(F.batch_norm(F.conv2d(xr, w3r, None, stride=2, padding=1), rm, rv, gr, br, training=True))
    y1 = F.conv2d(xr, w1r, None, stride=2)
    ((y3 * gy3).sum() + (y1 * gy1).sum()).backward()

    conv3 = torch.nn.Conv2d(C, C, 3, stride=2, padding=1, bias=False).to(DEV)
    conv1 = torch.nn.Conv2d(C, Co, 1, stride=2, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        conv3.weight.copy_(w3.to(DEV)); conv1.weight.copy_(w1.to(DEV)); bn.weight.copy_(gamma.to(DEV)); bn.bias.copy_(beta.to(DEV))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype).requires_grad_(True)
    h = nnops.conv_bn_act(xd, conv3, None, ACT_NONE, True, 0, dtype)
    h = nnops.BNAct.apply(nnops.Subsample2.apply(h), bn.weight, bn.bias, bn, True, 1, ACT_RELU)
    s = nnops.conv_bn_act(nnops.Subsample2.apply(xd), conv1, None, ACT_NONE, True, 0, dtype)
    ((h.float() * gy3.permute(0, 2, 3, 1).to(DEV)).sum() + (s.float() * gy1.permute(0, 2, 3, 1).to(DEV)).sum()).backward()
    # bf16: the conv output is STORED rounded, so ~0.3 % of the ReLU inputs change sign against the fp32 reference;
    # a flipped element is wrong by its full size: L2 error ~ sqrt(0.003) = 5 % on everything upstream of the ReLU
    tol = _tol(dtype, 2e-5, 8e-2)
    assert rel_err(h.float().cpu().permute(0, 3, 1, 2), y3.detach()) < tol
    assert rel_err(s.float().cpu().permute(0, 3, 1, 2), y1.detach()) < tol
    assert rel_err(xd.grad.float().cpu().permute(0, 3, 1, 2), xr.grad) < tol
    assert rel_err(conv3.weight.grad.cpu(), w3r.grad) < tol and rel_err(conv1.weight.grad.cpu(), w1r.grad) < tol
    assert rel_err(bn.weight.grad.cpu(), gr.grad) < tol and rel_err(bn.bias.grad.cpu(), br.grad) < tol
    assert rel_err(bn.running_mean.cpu(), rm) < tol and rel_err(bn.running_var.cpu(), rv) < tol


# ---------------------------------------------------------------------------------------------------------------
def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def build(pai, mults, patch, loss_type, seed, dtype=torch.float32, dropout=0.0):
    from thesis_pai_reconstruction_amd.models.trans_unet import TransUnetGAN
    m = TransUnetGAN(in_channels=1, out_channels=1, channel_mults=tuple(mults), patch_size=patch, dropout=dropout,
                     loss_type=loss_type)
    g = oracle.init_trans_state_portable(oracle.make_trans_unet_state(1, 1, tuple(mults), patch), seed)
    m.unet.load_state_dict(g, strict=True)
    d = None
    if loss_type == "gan":
        d = oracle.init_state_portable(oracle.make_disc_state(1), seed + 1)
        m.discriminator.load_state_dict(d, strict=True)
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m, g, d


def test_state_dict_keys_match_reference_layout(pai):
    from thesis_pai_reconstruction_amd.models.trans_unet import TransUnet
    for mults, patch in (((1, 1, 1, 2, 2), 2), ((1, 1, 1, 1, 1), 4)):
        u = TransUnet(1, 1, 256, mults, patch, 8, 0.0)
        want = oracle.make_trans_unet_state(1, 1, mults, patch)
        got = u.state_dict()
        assert set(got) == set(want)
        assert all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)


@pytest.mark.parametrize("name", ["ref_trans2_forward", "ref_trans4_forward",
                                  # BASELINE configs[4] as main.py:93-101 builds it: (1,2,2,4,4), patch_size 4, 1.03 B parameters
                                  "ref_trans4_forward_full"])
def test_forward_matches_reference_fixture(pai, golden_dir, name):
    z = _load(golden_dir, name)
    seed, n, size, fam = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), str(z["meta.family"])
    m, g, _ = build(pai, [int(v) for v in z["meta.mults"]], int(fam[5:]), "gan", seed)
    x, t = synth_batch(seed + 100, n, size)
    xd = x.to(DEV)
    m.unet.debug_capture = {}
    with torch.no_grad():
        pred = m.unet(xd)
        lf = m.discriminator(xd, pred)
    cap, m.unet.debug_capture = m.unet.debug_capture, None
    with torch.no_grad():
        _, acts = oracle.trans_unet_forward(g, x, training=True, return_feats=True)
    worst = {}
    for k, a in acts.items():
        got = cap[k].cpu()
        got = got.permute(0, 3, 1, 2) if got.dim() == 4 else got.view(a.shape)
        worst[k] = rel_err(got, a)
    assert max(worst.values()) < 1e-4, worst
    want = torch.from_numpy(z["pred_full"])
    assert float((pred.cpu() - want).abs().max()) < 1e-4 * float(want.abs().max())
    w = torch.from_numpy(z["logits_fake_full"])
    assert float((lf.cpu() - w).norm()) < 1e-4 * max(float(w.norm()), 1e-2)
    m.eval()                                   # eval mode (running statistics) against the live oracle
    gs = {k: v.detach().cpu().clone() for k, v in m.unet.state_dict().items()}
    with torch.no_grad():
        pe = m.unet(xd)
        we = oracle.trans_unet_forward(gs, x, training=False)
    assert float((pe.cpu() - we).abs().max()) < 1e-4 * float(we.abs().max())


def _check_step(m, z, s, gtol):
    for k, v in m.logged.items():
        want = float(z[f"step{s}.log.{k}"])
        assert abs(float(v) - want) <= (1e-4 if s == 0 else 1e-3) * max(1.0, abs(want)), (s, k, float(v), want)
    names = [k for k, _ in m.unet.named_parameters()]
    gmax = max(float(z[f"step{s}.ggrad.{k}"][3]) for k in names)
    bad = []
    for k, p in m.unet.named_parameters():
        want = z[f"step{s}.ggrad.{k}"]
        if p.dim() == 1 and float(want[3]) < 1e-4 * gmax:
            # analytically zero gradients (conv bias in front of a BatchNorm): cancellation noise in the reference
            assert p.grad is None or float(p.grad.abs().max()) < 1e-3 * gmax, (s, k)
            continue
        ok, worst = fingerprint_close(fingerprint(p.grad), want, gtol)
        if not ok:
            bad.append((k, worst))
    assert not bad, (s, bad[:6])


@pytest.mark.parametrize("name", ["ref_trans2_gan", "ref_trans2_ssim", "ref_trans4_gan_full"])
def test_training_step_matches_reference_fixture(pai, golden_dir, name):
    z = _load(golden_dir, name)
    seed, n, size, steps = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), int(z["meta.steps"])
    fam, loss_type = str(z["meta.family"]), str(z["meta.loss_type"])
    m, g, d = build(pai, [int(v) for v in z["meta.mults"]], int(fam[5:]), loss_type, seed)
    x, t = synth_batch(seed + 100, n, size)
    batch = (x.to(DEV), t.to(DEV))
    for s in range(steps):
        m.logged = {}
        m.training_step(batch, s)
        torch.cuda.synchronize()
        # Step 0 is the parity bar.  The oracle's own written-out attention (fp32, CPU) sits 3e-3 from these fixtures on
        # the gradients (tests/test_oracle_golden.py::test_trans_written_out_attention): 1e-7 of forward noise flips
        # ReLUs in the 4-16 M-element decoder tensors.  The same bound class applies here.
        _check_step(m, z, s, 1e-2 if s == 0 else 0.3 * s)
    for k, v in m.unet.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == (2 if loss_type == "gan" else 1) * steps
    m.eval()
    m.logged = {}
    with torch.no_grad():
        m.validation_step(batch, 0)
    for k, v in m.logged.items():
        want = float(z[f"val.log.{k}"])
        assert abs(float(v) - want) <= 2e-3 * max(1.0, abs(want)), (k, float(v), want)


def test_ragged_batch_and_bf16(pai, golden_dir):
    """A batch of 3 (token count 48: no power of two, the gather-GEMM fallbacks) in fp32 against the live oracle, then
    bf16 storage: finite, close on the output, and training makes progress."""
    z = _load(golden_dir, "ref_trans2_forward")
    seed = int(z["meta.seed"])
    mults = [int(v) for v in z["meta.mults"]]
    m, g, _ = build(pai, mults, 2, "gan", seed)
    x, t = synth_batch(seed + 300, 3, 256)
    with torch.no_grad():
        pred = m.unet(x.to(DEV))
        want = oracle.trans_unet_forward({k: v.clone() for k, v in g.items()}, x, training=True)
    assert float((pred.cpu() - want).abs().max()) < 1e-4 * float(want.abs().max())
    m16, _, _ = build(pai, mults, 2, "gan", seed, dtype=torch.bfloat16)
    x4, t4 = synth_batch(seed + 100, 4, 256)
    batch = (x4.to(DEV), t4.to(DEV))
    with torch.no_grad():
        p16 = m16.unet(batch[0])
    w4 = torch.from_numpy(z["pred_full"])
    # rounding-noise dominated at random initialisation (0.54 at the configs[4] size): sanity bound only; what is held
    # tightly is the training trajectory against the fp32 parity path (tests/test_gpu_configs.py: <= 0.05 % at full size)
    assert float((p16.cpu() - w4).norm() / w4.norm()) < 0.25
    m32, _, _ = build(pai, mults, 2, "gan", seed)
    first = None
    for s in range(4):
        logs = []
        for mm in (m16, m32):
            mm.logged = {}
            mm.training_step(batch, s)
            logs.append({k: float(v) for k, v in mm.logged.items()})
        vals, vals32 = logs
        assert all(np.isfinite(v) for v in vals.values()), vals
        for k in ("loss", "d_loss", "train_rmse", "train_psnr"):
            assert abs(vals[k] - vals32[k]) <= 0.02 * max(abs(vals32[k]), 1.0), (s, k, vals[k], vals32[k])
        first = first or vals
    assert vals["loss"] < first["loss"] and vals["train_rmse"] < first["train_rmse"]


def test_transformer_dropout_step_matches_reference_fixture(pai, golden_dir):
    """``dropout = 0.3`` against the REAL reference (tests/golden/ref_trans4_gan_dropout.npz, recorded with the global
    CPU generator seeded per step): the 96 masks of the step are re-drawn on the CPU from the same generator state by the
    oracle -- whose draws are pinned to the reference's by that same fixture, the attention operator's internal dropout
    included (tests/test_oracle_golden.py) -- and replayed into the HIP model's four Dropout sites per layer."""
    z = _load(golden_dir, "ref_trans4_gan_dropout")
    seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
    p = float(z["meta.dropout"])
    mults = [int(v) for v in z["meta.mults"]]
    m, g, d = build(pai, mults, 4, "gan", seed, dropout=p)
    x, t = synth_batch(seed + 100, n, size)
    torch.manual_seed(1000)               # step 0 of the recording run
    mask_log = []
    g0 = {k: v.clone() for k, v in g.items()}
    d0 = {k: v.clone() for k, v in d.items()}
    oracle.gan_training_step(g0, d0, oracle.AdamState(), oracle.AdamState(), x, t, dropout=p, mask_log=mask_log)
    assert len(mask_log) == 2 * 12 * 4
    queue = list(mask_log)

    def replay(site, li, shape, rate, device):
        s_, l_, mk = queue.pop(0)
        assert (s_, l_) == (site, li) and rate == p and mk.numel() == int(np.prod(shape)), (s_, l_, site, li)
        return mk.reshape(shape).to(device)

    m.unet.vit_bottleneck.dropout_mask_fn = replay
    m.logged = {}
    m.training_step((x.to(DEV), t.to(DEV)), 0)
    torch.cuda.synchronize()
    assert not queue
    _check_step(m, z, 0, 1e-2)


def test_transformer_dropout_step_matches_oracle_with_replayed_masks(pai):
    """Dropout(p) at the four sites of every encoder layer (attention weights inside pai_mha_*, behind the attention block,
    inside and behind the feed-forward block): one GAN step (two generator forwards, 96 masks) against the live oracle
    with the oracle's masks replayed into the HIP model, then a step on the model's own device-side draws."""
    mults, patch, p, seed = (1, 1, 1, 1, 1), 4, 0.3, 261
    m, g, d = build(pai, mults, patch, "gan", seed, dropout=p)
    assert not m.unet.supports_forward_reuse
    x, t = synth_batch(seed + 100, 3, 256)
    torch.manual_seed(5)
    mask_log = []
    logs, grads = oracle.gan_training_step(g, d, oracle.AdamState(), oracle.AdamState(), x, t, dropout=p,
                                           mask_log=mask_log, return_grads=True)
    assert len(mask_log) == 2 * 12 * 4
    queue = list(mask_log)

    def replay(site, li, shape, rate, device):
        s_, l_, mk = queue.pop(0)
        assert (s_, l_) == (site, li) and rate == p and mk.numel() == int(np.prod(shape)), (s_, l_, site, li)
        return mk.reshape(shape).to(device)

    m.unet.vit_bottleneck.dropout_mask_fn = replay
    m.logged = {}
    batch = (x.to(DEV), t.to(DEV))
    m.training_step(batch, 0)
    torch.cuda.synchronize()
    assert not queue
    for k, v in m.logged.items():
        assert abs(float(v) - float(logs[k])) <= 1e-4 * max(1.0, abs(float(logs[k]))), (k, float(v), float(logs[k]))
    gmax = max(float(v.norm()) for v in grads["g"].values())
    bad = {}
    for k, prm in m.unet.named_parameters():
        want = grads["g"][k]
        if float(want.norm()) < 1e-4 * gmax:
            continue
        e = rel_err(prm.grad.cpu(), want)
        if e > 1e-2:       # ReLU flips in the multi-million-element decoder tensors, as in the dropout-free step
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]
    m.unet.vit_bottleneck.dropout_mask_fn = None        # device-side draws: finite, and different from step to step
    vals = []
    for s in range(2):
        m.logged = {}
        m.training_step(batch, s + 1)
        vals.append({k: float(v) for k, v in m.logged.items()})
        assert all(np.isfinite(v) for v in vals[-1].values()), vals[-1]
