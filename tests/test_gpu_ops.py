"""GPU parity of the non-convolution entry points: BatchNorm (train/eval, forward/backward),
losses, tanh backward, denormalize, SSIM/PSNR/RMSE (forward and gradient), Adam, casts.
Checked against plain PyTorch-CPU fp32 ops and against the oracle's SSIM restatement
(itself pinned to scikit-image)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from _gpu_util import dev, from_nhwc, nhwc, q, rel_err, rnd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 6, 6), (2, 128, 16, 16), (5, 512, 2, 2)])
def test_batchnorm_forward_backward(pai, dtype, shape):
    from thesis_pai_reconstruction_amd import ops
    N, C, H, W = shape
    M = N * H * W
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    z = q(rnd(shape, 1) * 1.7 + 0.3, dtype)
    gamma, beta = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    rm0, rv0 = 0.05 * rnd((C,), 4), 1 + 0.2 * torch.rand(C)
    # reference: two running-stat updates on the same batch (GAN step runs G twice)
    rm, rv = rm0.clone(), rv0.clone()
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = F.batch_norm(zr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    F.batch_norm(z, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    a = F.leaky_relu(u, 0.2)
    g1, g2 = q(rnd(shape, 5), dtype), q(rnd(shape, 6), dtype)
    # two consumers: LeakyReLU path (g1) and ReLU skip path (g2)
    (F.leaky_relu(u, 0.2) * g1 + F.relu(u) * g2).sum().backward()

    # device side: statistics come as per-tile partials from the conv epilogue -> emulate 3 rows
    zf = z.permute(0, 2, 3, 1).reshape(M, C).double()
    parts = torch.zeros(3, 2, C, dtype=torch.float64)
    for r, chunk in enumerate(torch.chunk(zf, 3, dim=0)):
        parts[r, 0], parts[r, 1] = chunk.sum(0), (chunk * chunk).sum(0)
    stats = torch.zeros(ops.bn_stats_buffer_rows(3) * 2 * C, dtype=torch.float32, device=dev())
    stats[:3 * 2 * C] = parts.float().reshape(-1).to(dev())
    G, B = gamma.to(dev()), beta.to(dev())
    RM, RV = rm0.to(dev()), rv0.to(dev())
    nbt = torch.zeros((), dtype=torch.int64, device=dev())
    mean, rstd, scale, shift = (torch.empty(C, device=dev()) for _ in range(4))
    ops.bn_finalize(stats, 3, C, M, G, B, 1e-5, 0.1, 2, RM, RV, nbt, mean, rstd, scale, shift)
    Z = nhwc(z, dtype)
    A = torch.empty_like(Z)
    ops.bn_apply(dtype, Z, M, C, scale, shift, ops.ACT_LRELU, A)
    torch.cuda.synchronize()
    assert int(nbt) == 2
    assert rel_err(RM.cpu(), rm) < 1e-5 and rel_err(RV.cpu(), rv) < 1e-5
    assert rel_err(from_nhwc(A, N, H, W, C), a.detach()) < tol

    du = torch.empty_like(Z)
    dz = torch.empty_like(Z)
    partials = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C, device=dev())
    sums = torch.empty(2 * C, device=dev())
    dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    ops.bn_bwd_reduce(dtype, nhwc(g1, dtype), ops.ACT_LRELU, nhwc(g2, dtype), ops.ACT_RELU, A, Z, M, C, mean,
                      rstd, du, partials, sums, dg, db)
    ops.bn_bwd_apply(dtype, du, Z, M, C, mean, rstd, G, sums, dz)
    torch.cuda.synchronize()
    assert rel_err(dg.cpu(), gr.grad) < max(tol, 2e-4)
    assert rel_err(db.cpu(), br.grad) < max(tol, 2e-4)
    assert rel_err(from_nhwc(dz, N, H, W, C), zr.grad) < max(tol, 2e-4)

    # single-consumer forms of the composable networks: pass 1 stores du (6 tensor passes) or nothing (5: pass 2 rebuilds
    # du from g1 and z) -- the same dz, dgamma, dbeta bit for bit
    for act in (ops.ACT_LRELU, ops.ACT_RELU):
        G1 = nhwc(g1, dtype)
        du6, dz6, dz5 = torch.empty_like(Z), torch.empty_like(Z), torch.empty_like(Z)
        dg6, db6, dg5, db5 = (torch.zeros(C, device=dev()) for _ in range(4))
        ops.bn_bwd_reduce_affine(dtype, G1, act, None, ops.ACT_NONE, Z, M, C, scale, shift, mean, rstd, du6, partials, sums, dg6, db6)
        ops.bn_bwd_apply(dtype, du6, Z, M, C, mean, rstd, G, sums, dz6)
        ops.bn_bwd_reduce_affine(dtype, G1, act, None, ops.ACT_NONE, Z, M, C, scale, shift, mean, rstd, None, partials, sums, dg5, db5)
        ops.bn_bwd_apply_affine(dtype, G1, act, Z, M, C, scale, shift, mean, rstd, G, sums, dz5)
        torch.cuda.synchronize()
        assert torch.equal(dz5, dz6) and torch.equal(dg5, dg6) and torch.equal(db5, db6), act
    with pytest.raises(ops.PaiError):      # a second gradient needs the stored du
        ops.bn_bwd_reduce_affine(dtype, G1, ops.ACT_RELU, G1, ops.ACT_RELU, Z, M, C, scale, shift, mean, rstd, None, partials,
                                 sums, dg5, db5)

    # eval mode
    ops.bn_eval_coeffs(C, G, B, RM, RV, 1e-5, scale, shift)
    ops.bn_apply(dtype, Z, M, C, scale, shift, ops.ACT_NONE, A)
    want = F.batch_norm(z, rm, rv, gamma, beta, training=False, eps=1e-5)
    assert rel_err(from_nhwc(A, N, H, W, C), want) < tol


@pytest.mark.parametrize("C,R", [(64, 300), (128, 2049), (128, 32768), (64, 10007), (8, 5000)], ids=str)
def test_bn_finalize_many_partial_rows(pai, C, R):
    """More than 16 partial rows: the wide fp64 tree; more than 2048 (the 1 x 1 convolutions of the residual U-Net at
    512 x 512 write 32768): chunk sums over the whole chip first, in the scratch rows behind the partials."""
    from thesis_pai_reconstruction_amd import ops
    M = R * 128
    rng = np.random.default_rng(0)
    parts = rng.standard_normal((R, 2, C)).astype(np.float32)
    parts[:, 1] = np.abs(parts[:, 1]) * 200 + 300
    stats = torch.zeros(ops.bn_stats_buffer_rows(R) * 2 * C, dtype=torch.float32, device=dev())
    stats[:R * 2 * C] = torch.from_numpy(parts).reshape(-1).to(dev())
    mean, rstd, scale, shift = (torch.empty(C, device=dev()) for _ in range(4))
    ops.bn_finalize(stats, R, C, M, None, None, 1e-5, 0.1, 1, None, None, None, mean, rstd, scale, shift)
    s = parts.astype(np.float64).sum(0)
    m = s[0] / M
    v = s[1] / M - m * m
    np.testing.assert_allclose(mean.cpu().numpy(), m, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(rstd.cpu().numpy(), 1 / np.sqrt(v + 1e-5), rtol=1e-6)


def test_losses_and_head_backward(pai):
    from thesis_pai_reconstruction_amd import functional as PF
    logits = rnd((4, 1, 15, 15), 1, 2.0)
    for tgt in (0.0, 1.0):
        lr = logits.clone().requires_grad_(True)
        want = F.binary_cross_entropy_with_logits(lr, torch.full_like(lr, tgt))
        want.backward()
        lg = logits.to(dev()).requires_grad_(True)
        got = PF.bce_with_logits_const(lg, tgt)
        (got * 1.0).backward()
        assert abs(float(got) - float(want)) < 1e-6
        assert rel_err(lg.grad.cpu(), lr.grad) < 1e-5
    p, t = rnd((3, 1, 64, 64), 2), rnd((3, 1, 64, 64), 3)
    for fn, ref in ((PF.l1_loss, F.l1_loss), (PF.mse_loss, F.mse_loss)):
        pr = p.clone().requires_grad_(True)
        want = ref(pr, t)
        (50 * want).backward()
        pg = p.to(dev()).requires_grad_(True)
        got = fn(pg, t.to(dev()))
        (50 * got).backward()
        assert abs(float(got) - float(want)) < 1e-6
        assert rel_err(pg.grad.cpu(), pr.grad) < 1e-6
    x = rnd((2, 1, 32, 32), 4, 1.5)
    xr = x.clone().requires_grad_(True)
    oracle.denormalize(xr).pow(2).sum().backward()
    xg = x.to(dev()).requires_grad_(True)
    PF.denormalize(xg).pow(2).sum().backward()
    assert torch.equal(PF.denormalize(x.to(dev())).cpu(), oracle.denormalize(x))
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("shape", [(4, 1, 256, 256), (3, 1, 64, 48), (2, 3, 32, 32), (4, 1, 16, 256)])
def test_ssim_psnr_rmse(pai, shape):
    from thesis_pai_reconstruction_amd import functional as PF
    rng = np.random.default_rng(shape[2])
    a = torch.from_numpy(rng.random(shape, dtype=np.float32))
    b = torch.clamp(a + 0.1 * torch.from_numpy(rng.standard_normal(shape).astype(np.float32)), 0, 1)
    A, B = a.to(dev()), b.to(dev())
    per, full = oracle.ssim_full(b, a)
    gper, gfull = PF.ssim_per_image(B, A, return_full_image=True)
    assert float((gper.cpu() - per).abs().max()) < 5e-6
    # the per-image ORDERING is part of the parity criterion (north_star)
    assert torch.equal(torch.argsort(gper.cpu()), torch.argsort(per))
    assert float((gfull.cpu() - full).abs().max()) < 2e-5
    assert abs(float(PF.ssim(B, A)) - float(oracle.ssim(b, a))) < 2e-6
    assert abs(float(PF.psnr(B, A)) - float(oracle.psnr(b, a))) < 2e-5
    assert abs(float(PF.rmse(B, A)) - float(oracle.rmse(b, a))) < 1e-7
    # fused denormalisation
    x, y = a * 2.4 - 1.2, b * 2.4 - 1.2
    s, p, r = PF.metrics_of_normalized(x.to(dev()), y.to(dev()))
    dx, dy = oracle.denormalize(x), oracle.denormalize(y)
    assert abs(float(s) - float(oracle.ssim(dx, dy))) < 2e-6
    assert abs(float(p) - float(oracle.psnr(dx, dy))) < 2e-5
    assert abs(float(r) - float(oracle.rmse(dx, dy))) < 1e-7


@pytest.mark.parametrize("weights", [(1.0, 0.0), (0.0, 1.0), (30.0, 1.0)])
def test_ssim_psnr_loss_gradient(pai, weights):
    from thesis_pai_reconstruction_amd import functional as PF
    ws, wp = weights
    rng = np.random.default_rng(7)
    x = torch.from_numpy((rng.random((2, 1, 48, 40), dtype=np.float32) * 2.6 - 1.3))
    t = torch.from_numpy((rng.random((2, 1, 48, 40), dtype=np.float32) * 2 - 1))
    xr = x.clone().requires_grad_(True)
    dp, dt = oracle.denormalize(xr), oracle.denormalize(t)
    want = -(ws * oracle.ssim(dp, dt) + wp * oracle.psnr(dp, dt))
    want.backward()
    xg = x.to(dev()).requires_grad_(True)
    got = -PF.ssim_psnr_of_normalized(xg, t.to(dev()), ws, wp)
    got.backward()
    assert abs(float(got) - float(want)) < 2e-5 * max(1.0, abs(float(want)))
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-4


def test_adam_matches_torch(pai):
    from thesis_pai_reconstruction_amd import ops
    p0, steps = rnd((10007,), 1), 3
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
    P = p0.to(dev())
    m, v = torch.zeros_like(P), torch.zeros_like(P)
    for s in range(1, steps + 1):
        g = rnd((10007,), 10 + s, 0.01)
        ref.grad = g.clone()
        opt.step()
        ops.adam(P, g.to(dev()), m, v, 2e-4, 0.5, 0.999, 1e-7, s)
    assert float((P.cpu() - ref.detach()).abs().max()) < 1e-7


@pytest.mark.parametrize("cout,taps,cin,pre,post", [(128, 16, 64, 192, 384), (64, 4, 192, 0, 0), (64, 16, 64, 64, 5)])
def test_adam_pack_equals_adam_then_pack(pai, cout, taps, cin, pre, post):
    """pai_adam_pack (the streamed optimizer step that writes a dense layer's bf16 filter packs from the block that
    produced the new weights) against pai_adam followed by pai_pack_weights, bit for bit: parameters, both moments,
    both packs, with neighbours in front of and behind the weight inside the range."""
    from thesis_pai_reconstruction_amd import ops
    wn = cout * taps * cin
    n = pre + wn + post
    p0, g = rnd((n,), 3).to(dev()), rnd((n,), 4, 0.01).to(dev())
    m0, v0 = rnd((n,), 5, 0.01).to(dev()), rnd((n,), 6, 0.01).abs().to(dev())
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    bf = torch.bfloat16
    wfa, wda = torch.empty(wn, dtype=bf, device=dev()), torch.empty(wn, dtype=bf, device=dev())
    wfb, wdb = torch.full((wn,), 7.0, dtype=bf, device=dev()), torch.full((wn,), 7.0, dtype=bf, device=dev())
    for step in (1, 2, 5):
        ops.adam(pa, g, ma, va, 2e-4, 0.5, 0.999, 1e-7, step)
        ops.pack_weights(bf, pa[pre:pre + wn], cout, taps, cin, wfa, wda)
        ops.adam_pack(pb, g, mb, vb, pre, cout, taps, cin, wfb, wdb, 2e-4, 0.5, 0.999, 1e-7, step)
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), step
        assert torch.equal(wfa, wfb) and torch.equal(wda, wdb), step
    # one pack only
    wdb.fill_(7.0)
    ops.adam_pack(pb, g, mb, vb, pre, cout, taps, cin, wfb, None, 2e-4, 0.5, 0.999, 1e-7, 6)
    assert float(wdb.float().min()) == 7.0
    with pytest.raises(ops.PaiError):
        ops.adam_pack(pb, g, mb, vb, pre, cout + 1, taps, cin, wfb, wdb, 2e-4, 0.5, 0.999, 1e-7, 1)
    with pytest.raises(ops.PaiError):
        ops.adam_pack(pb, g, mb, vb, pre + post + 64, cout, taps, cin, wfb, wdb, 2e-4, 0.5, 0.999, 1e-7, 1)


def test_cast_roundtrip(pai):
    from thesis_pai_reconstruction_amd import ops
    x = rnd((1000,), 1).to(dev())
    b = torch.empty(1000, dtype=torch.bfloat16, device=dev())
    ops.cast(x, b)
    assert torch.equal(b.cpu(), x.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (3, 128, 4, 4), (1, 8, 2, 2)])
def test_maxpool_upsample_add(pai, dtype, shape):
    """nn.MaxPool2d(2), nearest nn.Upsample(scale_factor=2) and the residual sum (+ReLU) of the residual U-Net
    (reference models/res_unet.py:199,231,74) against torch, forward and backward; exact in both dtypes (pure
    selection / replication; the sums are compared to bf16 rounding)."""
    from thesis_pai_reconstruction_amd import ops
    N, C, H, W = shape
    x = q(rnd(shape, 1), dtype).requires_grad_(True)
    # ---- max pool ----
    y = F.max_pool2d(x, 2)
    gy = q(rnd(tuple(y.shape), 2), dtype)
    y.backward(gy)
    X = nhwc(x.detach(), dtype)
    out = torch.empty(N * (H // 2) * (W // 2) * C, dtype=dtype, device=dev())
    idx = torch.empty(out.numel(), dtype=torch.uint8, device=dev())
    ops.maxpool2(dtype, X, N, H, W, C, out, idx)
    dx = torch.empty_like(X)
    ops.maxpool2_bwd(dtype, nhwc(gy, dtype), idx, N, H, W, C, dx)
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(out, N, H // 2, W // 2, C), y.detach())
    assert torch.equal(from_nhwc(dx, N, H, W, C), x.grad)
    # ---- nearest upsample ----
    x.grad = None
    u = F.interpolate(x, scale_factor=2)              # nn.Upsample default mode='nearest'
    gu = q(rnd(tuple(u.shape), 3), dtype)
    u.backward(gu)
    up = torch.empty(N * 2 * H * 2 * W * C, dtype=dtype, device=dev())
    ops.upsample2(dtype, X, N, H, W, C, up)
    dxu = torch.empty_like(X)
    ops.upsample2_bwd(dtype, nhwc(gu, dtype), N, H, W, C, dxu)
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(up, N, 2 * H, 2 * W, C), u.detach())
    assert rel_err(from_nhwc(dxu, N, H, W, C), x.grad) < (1e-6 if dtype == torch.float32 else 4e-3)
    # ---- residual sum + ReLU; backward = activation backward on the stored sum ----
    b = q(rnd(shape, 4), dtype)
    s = torch.empty_like(X)
    ops.add_act(dtype, X, nhwc(b, dtype), ops.ACT_RELU, s)
    want = F.relu(x.detach() + b)
    assert rel_err(from_nhwc(s, N, H, W, C), want) < (1e-6 if dtype == torch.float32 else 4e-3)
    g = nhwc(q(rnd(shape, 5), dtype), dtype)
    ds = torch.empty_like(X)
    ops.act_bwd(dtype, g, ops.ACT_RELU, None, ops.ACT_NONE, s, s.numel(), ds)
    torch.cuda.synchronize()
    mask = (from_nhwc(s, N, H, W, C) > 0).float()
    assert torch.equal(from_nhwc(ds, N, H, W, C), from_nhwc(g, N, H, W, C) * mask)


def test_multi_adam_equals_stock_adam(pai):
    """MultiAdam (pai_adam_multi: separately allocated parameters, chunks of 48 tensors per launch) against
    torch.optim.Adam over three steps, 120 tensors of mixed sizes; state_dict keeps the stock layout."""
    from thesis_pai_reconstruction_amd.optim import MultiAdam
    rng = np.random.default_rng(5)
    shapes = [(int(rng.integers(1, 70)), int(rng.integers(1, 50))) for _ in range(117)] + [(1,), (300000,), (1025, 513)]
    pa = [torch.nn.Parameter(rnd(s, 100 + i).to(dev())) for i, s in enumerate(shapes)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = MultiAdam(pa, lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
    ob = torch.optim.Adam(pb, lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
    for step in range(3):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = (rnd(a.shape, 1000 * step + i) * 0.1).to(dev())
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 2e-7 + 1e-6 * float(b.abs().max())
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert len(sa) == len(sb) and float(sa[0]["step"]) == 3
    # the C ABI takes beta2 as a float: 1 - 0.999f differs from torch's float(1 - 0.999) by 1.3e-5 relative
    assert rel_err(sa[5]["exp_avg_sq"].cpu(), sb[5]["exp_avg_sq"].cpu()) < 3e-5


def test_comm_abi_single_rank(pai):
    """pai_comm_unique_id / pai_comm_init / pai_allreduce / pai_comm_destroy on a communicator of one rank (all a 1-GPU
    box can host): RCCL is found and loaded lazily, the in-place SUM over one rank leaves fp32 and bf16 buffers unchanged,
    and GradReducer accepts the communicator.  The multi-rank exchange itself is the driver's 8-GPU run."""
    from thesis_pai_reconstruction_amd import dist as pdist, ops
    uid = ops.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = ops.Comm(uid, 0, 1)
    for dtype in (torch.float32, torch.bfloat16):
        t = rnd((3, 1000), 77).to(dev()).to(dtype)
        want = t.clone()
        comm.all_reduce(t)
        torch.cuda.synchronize()
        assert torch.equal(t, want)
    red = pdist.GradReducer(comm=comm)
    work = red._all_reduce_async(t)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    comm.destroy()


def test_pack_weights_multi_equals_per_layer_packs(pai):
    """pai_pack_weights_multi (every 64-multiple layer of a network in one launch) writes the same bf16 forward and
    input-gradient packs as one pai_pack_weights launch per layer; NULL outputs are skipped."""
    from thesis_pai_reconstruction_amd import ops
    torch.manual_seed(5)
    shapes = [(128, 16, 64), (64, 16, 256), (512, 16, 512), (64, 9, 64)]        # (Cout, taps, Cin)
    items, want = [], []
    for i, (co, taps, ci) in enumerate(shapes):
        w = torch.randn(co * taps * ci, device=dev())
        wf, wd = (torch.empty(w.numel(), dtype=torch.bfloat16, device=dev()) for _ in range(2))
        ops.pack_weights(torch.bfloat16, w, co, taps, ci, wf, wd)
        mf = torch.full_like(wf, -7.0)
        md = None if i == 1 else torch.full_like(wd, -7.0)       # one layer without an input-gradient pack
        items.append((w, co, taps, ci, mf, md))
        want.append((wf, wd))
    ops.pack_weights_multi(items)
    for (w, co, taps, ci, mf, md), (wf, wd) in zip(items, want):
        assert torch.equal(mf, wf)
        if md is not None:
            assert torch.equal(md, wd)
    with pytest.raises(ops.PaiError, match="multiples of 64"):
        ops.pack_weights_multi([(torch.zeros(32 * 16 * 64, device=dev()), 32, 16, 64,
                                 torch.empty(32 * 16 * 64, dtype=torch.bfloat16, device=dev()), None)])


def test_fused_gan_losses_equal_the_composed_ones(pai):
    """gan_discriminator_loss_pairs / gan_generator_loss (two loss launches into one fp64 scalar + pai_scalar_take)
    against the composed bce / l1 functions the reference spells out (models/wrapper.py:44-50,68-95): values and
    gradients; metrics_of_normalized (pai_metrics_take) against ssim / psnr / rmse of the denormalised pair."""
    from thesis_pai_reconstruction_amd import functional as PF
    torch.manual_seed(2)
    n = 3
    labels = torch.randn(2 * n, 1, 6, 6, device=dev(), requires_grad=True)
    d1 = PF.gan_discriminator_loss_pairs(labels, n)
    (g1,) = torch.autograd.grad(d1 * 1.5, labels)
    l2 = labels.detach().clone().requires_grad_(True)
    d2 = PF.bce_with_logits_const(l2[n:], 0.0) + PF.bce_with_logits_const(l2[:n], 1.0)
    (g2,) = torch.autograd.grad(d2 * 1.5, l2)
    assert abs(float(d1) - float(d2)) <= 1e-6 * max(1.0, abs(float(d2))) and torch.allclose(g1, g2, rtol=1e-6, atol=1e-9)
    for _ in range(2):          # twice: the accumulator is re-armed by the take
        pl = torch.randn(n, 1, 6, 6, device=dev(), requires_grad=True)
        pred = torch.tanh(torch.randn(n, 1, 32, 32, device=dev())).requires_grad_(True)
        tgt = torch.tanh(torch.randn(n, 1, 32, 32, device=dev()))
        a = PF.gan_generator_loss(pl, pred, tgt, 100.0)
        ga = torch.autograd.grad(a, (pl, pred))
        b = PF.bce_with_logits_const(pl, 1.0) + 100.0 * PF.l1_loss(pred, tgt)
        gb = torch.autograd.grad(b, (pl, pred))
        assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(b)))
        assert torch.allclose(ga[0], gb[0], rtol=1e-6, atol=1e-9) and torch.allclose(ga[1], gb[1], rtol=1e-6, atol=1e-9)
        s, p, r = PF.metrics_of_normalized(pred.detach(), tgt)
        dp, dt = PF.denormalize(pred.detach()), PF.denormalize(tgt)
        assert abs(float(s) - float(PF.ssim(dp, dt))) <= 1e-6
        assert abs(float(p) - float(PF.psnr(dp, dt))) <= 1e-5 and abs(float(r) - float(PF.rmse(dp, dt))) <= 1e-7


@pytest.mark.parametrize("shape,groups", [((64, 32, 1, 1), 1), ((128, 4, 3, 3), 32), ((24, 7, 3, 3), 1), ((32, 8, 1, 1), 4),
                                          ((6, 5, 4, 4), 2)])
def test_filter_layout_kernels(pai, shape, groups):
    """pai_filter_to_dense / pai_filter_grad_from_dense against the permute / block-diagonal formulation."""
    from thesis_pai_reconstruction_amd import ops

    cout, cig, kh, kw = shape
    g = torch.Generator().manual_seed(3)
    w = torch.randn(*shape, generator=g).to(dev())
    dense = torch.full((cout, kh, kw, cig * groups), 7.0, device=dev())
    ops.filter_to_dense(w, cout, cig, kh * kw, groups, dense)
    want = torch.zeros(cout, kh, kw, cig * groups, device=dev())
    cog = cout // groups
    for gi in range(groups):
        want[gi * cog:(gi + 1) * cog, :, :, gi * cig:(gi + 1) * cig] = w[gi * cog:(gi + 1) * cog].permute(0, 2, 3, 1)
    assert torch.equal(dense, want)
    dw_dense = torch.randn(cout, kh, kw, cig * groups, generator=g).to(dev())
    dw = torch.full(shape, 7.0, device=dev())
    ops.filter_grad_from_dense(dw_dense, cout, cig, kh * kw, groups, dw)
    want = torch.empty(shape, device=dev())
    for gi in range(groups):
        want[gi * cog:(gi + 1) * cog] = dw_dense[gi * cog:(gi + 1) * cog, :, :, gi * cig:(gi + 1) * cig].permute(0, 3, 1, 2)
    assert torch.equal(dw, want)
    with pytest.raises(ops.PaiError):
        ops.filter_to_dense(w, cout, cig, kh * kw, 5 if cout % 5 else 7, dense)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("skip_bn", [True, False], ids=["conv_skip", "identity_skip"])
@pytest.mark.parametrize("act_a,act", [(0, 0), (2, 0), (0, 2)], ids=["plain", "relu_branch", "relu_sum"])
def test_bn2_add_act(pai, dtype, skip_bn, act_a, act):
    """pai_bn2_add_act against the three passes it replaces (bn_apply, bn_apply, add_act): fp32 bit for bit; bf16 against
    the fp32 formula (the fused pass rounds once, the three passes three times)."""
    from thesis_pai_reconstruction_amd import ops

    M, C = 3 * 7 * 5, 24
    g = torch.Generator().manual_seed(11)
    za, zb = (torch.randn(M, C, generator=g).to(dev()).to(dtype) for _ in range(2))
    sca, sha, scb, shb = (torch.randn(C, generator=g).to(dev()) for _ in range(4))
    out = torch.empty_like(za)
    ops.bn2_add_act(dtype, za, sca, sha, zb, scb if skip_bn else None, shb if skip_bn else None, M, C, act_a, act, out)
    f = {0: lambda t: t, 2: torch.relu}
    a = f[act_a](torch.addcmul(sha, za.float(), sca))
    b = torch.addcmul(shb, zb.float(), scb) if skip_bn else zb.float()
    want = f[act](a + b)
    if dtype == torch.float32:
        a3, b3 = torch.empty_like(za), torch.empty_like(zb)
        ops.bn_apply(dtype, za, M, C, sca, sha, act_a, a3)
        if skip_bn:
            ops.bn_apply(dtype, zb, M, C, scb, shb, 0, b3)
        else:
            b3 = zb
        three = torch.empty_like(za)
        ops.add_act(dtype, a3, b3, act, three)
        assert torch.equal(out, three)
        assert rel_err(out, want) < 1e-6
    else:
        assert rel_err(out.float(), want) < 4e-3
    with pytest.raises(ops.PaiError):
        ops.bn2_add_act(dtype, za, sca, sha, zb, scb, None, M, C, act_a, act, out)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_swap_mid_and_fork(pai, dtype):
    """pai_swap_mid = permute(0, 2, 1, 3) (its own inverse with the middle extents exchanged); nnops.Fork hands the sum of
    the two consumers' gradients to the producer."""
    from thesis_pai_reconstruction_amd import nnops, ops

    A, B, Cc, D = 6, 4, 3, 16
    x = torch.randn(A, B, Cc, D, device=dev()).to(dtype)
    y = torch.empty(A, Cc, B, D, device=dev(), dtype=dtype)
    ops.swap_mid(x, A, B, Cc, D, y)
    assert torch.equal(y, x.permute(0, 2, 1, 3).contiguous())
    back = torch.empty_like(x)
    ops.swap_mid(y, A, Cc, B, D, back)
    assert torch.equal(back, x)
    with pytest.raises(ops.PaiError):
        ops.swap_mid(x, A, B, Cc, D + 1, y)
    t = torch.randn(2, 5, 3, 8, device=dev()).to(dtype).requires_grad_(True)
    s = nnops.SwapMid.apply(t.view(10, 3, 8), 2, 5, 3, 8)
    w = torch.randn_like(s)
    (s.float() * w.float()).sum().backward()
    assert torch.equal(t.grad.view(2, 5, 3, 8), w.view(2, 3, 5, 8).permute(0, 2, 1, 3).contiguous().to(dtype))
    u = torch.randn(2, 4, 4, 8, device=dev()).to(dtype).requires_grad_(True)
    a, b = nnops.fork(u)
    wa, wb = torch.randn_like(u), torch.randn_like(u)
    ((a.float() * wa.float()).sum() + (b.float() * wb.float()).sum()).backward()
    want = (wa.float() + wb.float()).to(dtype) if dtype == torch.float32 else None
    if dtype == torch.float32:
        assert torch.equal(u.grad, want)
    else:
        assert rel_err(u.grad.float(), wa.float() + wb.float()) < 4e-3
