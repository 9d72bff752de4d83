"""GPU: the composite entry points pai_conv_fwd_bn / pai_conv_dgrad_bn_apply on the split-K layers of the U-Net bottleneck
-- convolution + BatchNorm2d(train) + activation, and input gradient + the producer's whole BatchNorm backward (reference
models/pix2pix.py:63-70,99-106; aten::native_batch_norm / native_batch_norm_backward behind nn.BatchNorm2d).

Against PyTorch-CPU fp32 autograd on integer inputs (the convolution part is exact, the BatchNorm part is held to fp32
rounding), at BASELINE configs[1] layer shapes.  (Rounds 3-4 also ran these layers through a single column-owner finish
launch, csrc/gg_finish.hip, tunable finish_fused; it was slower than the three launches it replaced -- 6.64-6.65 against
6.52-6.56 ms/step -- and was removed in round 5; pai_conv_bn_fused now always answers 0.)

Round 5: for these small layers BatchNorm finalize + apply (forward and backward) run as ONE launch
(bn_fin_apply_k / bn_bwd_fin_apply_k, tunable bn_fuse_small) that sums the partial rows in the order of the finalize
kernels: every output must be BIT-IDENTICAL to the two-launch form."""
import pytest
import torch
import torch.nn.functional as F

from _gpu_util import dev, from_nhwc, fwd_pack, nhwc

pytestmark = pytest.mark.gpu

# (name, transposed, N, H, C1, C2, Cout): output rows N * OH * OW <= 4096 and a long reduction -> split-K with the workspace
CASES = [
    ("enc_small", 0, 4, 8, 256, 0, 256),        # 4 x 4 x 4 = 64 rows
    ("dec_small", 1, 4, 4, 256, 256, 256),      # 4 phases x 64 rows
    ("cfg2_enc4", 0, 64, 16, 512, 0, 512),      # 64 x 8 x 8 = 4096 rows: the largest layer the finish takes
    ("cfg2_enc6", 0, 64, 4, 512, 0, 512),       # 256 rows
    ("cfg2_dec2", 1, 64, 4, 512, 512, 512),     # 4 x 1024 rows
]


def _ints(shape, seed, lo=-2, hi=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _setup(pai, case):
    from thesis_pai_reconstruction_amd import ops
    name, tr, N, H, C1, C2, Cout = case
    d = ops.make_desc(torch.bfloat16, tr, N, H, H, C1, C2, Cout, 2, 0, 0, ops.ACT_NONE)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    return ops, d


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bn_act_forward_fused_finish(pai, case):
    ops, d = _setup(pai, case)
    name, tr, N, H, C1, C2, Cout = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    M = N * OH * OH
    x1 = _ints((N, C1, H, H), 1)
    x2 = _ints((N, C2, H, H), 2) if C2 else None
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3)
    bias = _ints((Cout,), 7, -3, 3)
    gamma = torch.linspace(0.5, 1.5, Cout)
    beta = torch.linspace(-0.25, 0.25, Cout)
    x = torch.cat([x1] + ([x2] if C2 else []), 1)
    z_ref = (F.conv_transpose2d(x, w, bias, stride=2, padding=1) if tr else F.conv2d(x, w, bias, stride=2, padding=1))
    assert float(z_ref.abs().max()) < 2 ** 24
    zb = z_ref.bfloat16().float()                      # what is stored, and what the normalisation is applied to
    mean_ref = z_ref.double().mean((0, 2, 3))
    var_ref = z_ref.double().var((0, 2, 3), unbiased=False)
    rstd_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
    a_ref = F.leaky_relu((zb.double() - mean_ref.view(1, -1, 1, 1)) * (rstd_ref * gamma.double()).view(1, -1, 1, 1)
                         + beta.double().view(1, -1, 1, 1), 0.2).float()

    wm = fwd_pack(w, bool(tr))
    wf = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, Cin, wf, None)
    X1, X2 = nhwc(x1, dt), (nhwc(x2, dt) if C2 else None)
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, device=dev())
    out = {}
    for fused in (1, 0):
        ops.set_tunable("bn_fuse_small", fused)
        try:
            assert ops.conv_bn_fused(d, 0) is False
            z = torch.empty(M * Cout, dtype=dt, device=dev())
            a = torch.empty_like(z)
            rm, rv = torch.zeros(Cout, device=dev()), torch.ones(Cout, device=dev())
            nbt = torch.zeros((), dtype=torch.int64, device=dev())
            st = [torch.empty(Cout, device=dev()) for _ in range(4)]
            ops.conv_fwd_bn(d, X1, X2, wf, bias.to(dev()), z, a, ops.ACT_LRELU, gamma.to(dev()), beta.to(dev()), 1e-5, 0.1, 2,
                            rm, rv, nbt, st[0], st[1], st[2], st[3], stats)
            torch.cuda.synchronize()
            out[fused] = (from_nhwc(z, N, OH, OH, Cout), from_nhwc(a, N, OH, OH, Cout), [t.cpu() for t in st], rm.cpu(), rv.cpu(), int(nbt))
        finally:
            ops.set_tunable("bn_fuse_small")
    for fused, (z, a, st, rm, rv, nbt) in out.items():
        assert torch.equal(z, zb), (name, fused, "z")                       # integer data: exact
        assert torch.allclose(st[0].double(), mean_ref, rtol=1e-6, atol=1e-6), (name, fused, "mean")
        assert torch.allclose(st[1].double(), rstd_ref, rtol=1e-5), (name, fused, "rstd")
        # a: bf16 of an fp32 affine of exact inputs -- one bf16 step of slack for the rounding of scale / shift
        assert float((a - a_ref).abs().max()) <= 2.0 ** -7 * max(1.0, float(a_ref.abs().max())), (name, fused, "a")
        assert nbt == 2
        unb = var_ref * M / (M - 1)
        rm_ref, rv_ref = torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64)
        for _ in range(2):
            rm_ref = 0.9 * rm_ref + 0.1 * mean_ref
            rv_ref = 0.9 * rv_ref + 0.1 * unb
        assert torch.allclose(rm.double(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv.double(), rv_ref, rtol=1e-5), (name, fused)
    # one launch against two: the same summation order, so everything is bit-identical
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]), name
    for k in range(4):
        assert torch.equal(out[1][2][k], out[0][2][k]), (name, k)
    assert torch.equal(out[1][3], out[0][3]) and torch.equal(out[1][4], out[0][4]) and out[1][5] == out[0][5], name


@pytest.mark.parametrize("enc_form", [False, True], ids=["decoder_form", "encoder_form"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_dgrad_with_producer_batchnorm_backward_fused_finish(pai, case, enc_form):
    """dz of the PRODUCER layer from the consumer's input gradient: du = act1'(pre) g (+ act2'(pre) add), then the
    BatchNorm backward dz = gamma rstd (du - mean(du) - xhat mean(du xhat)), dgamma, dbeta -- against PyTorch-CPU fp32
    autograd of  y = conv(act(BN(z)))  (+ the skip consumer), and fused against the three-launch path."""
    ops, d = _setup(pai, case)
    name, tr, N, H, C1, C2, Cout = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    M = N * H * H
    if M > 4096:
        pytest.skip("the producer tensor has more than 4096 rows: not a fused-finish layer")
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3)
    dy = _ints((N, Cout, OH, OH), 5)
    z = _ints((N, C1, H, H), 11, -3, 3)                  # producer's convolution output
    gskip = _ints((N, C1, H, H), 12) if enc_form else None
    gamma = torch.linspace(0.5, 1.5, C1)
    beta = torch.linspace(-0.4, 0.4, C1)
    act1 = ops.ACT_LRELU if enc_form else ops.ACT_RELU
    # reference: z -> BN(train) -> act1 -> (this layer's input x1) -> conv -> <y, dy>;  + <relu(BN(z)), gskip> for the skip
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    bn = F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5)
    x1 = F.leaky_relu(bn, 0.2) if enc_form else F.relu(bn)
    x = torch.cat([x1, torch.zeros(N, C2, H, H)], 1) if C2 else x1
    y = F.conv_transpose2d(x, w, None, stride=2, padding=1) if tr else F.conv2d(x, w, None, stride=2, padding=1)
    loss = (y * dy).sum()
    if enc_form:
        loss = loss + (F.relu(bn) * gskip).sum()
    loss.backward()
    mean = z.double().mean((0, 2, 3))
    var = z.double().var((0, 2, 3), unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5))
    scale, shift = (gamma.double() * rstd).float(), (beta.double() - mean * gamma.double() * rstd).float()

    wm = fwd_pack(w, bool(tr))
    wd = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, Cin, None, wd)
    f = lambda t: None if t is None else t.to(dev())
    part = torch.empty(ops.conv_dgrad_bn_rows_max(d) * 2 * C1, device=dev())
    res = {}
    for fused in (1, 0):
        ops.set_tunable("bn_fuse_small", fused)
        try:
            assert ops.conv_bn_fused(d, 1) is False
            du = torch.empty(M * C1, dtype=dt, device=dev())
            dx2 = torch.empty(M * C2, dtype=dt, device=dev()) if C2 else None
            dz = torch.empty(M * C1, dtype=dt, device=dev())
            sums = torch.empty(2 * C1, device=dev())
            dgamma, dbeta = torch.full((C1,), 1.0, device=dev()), torch.full((C1,), -1.0, device=dev())   # += semantics
            ops.conv_dgrad_bn_apply(d, nhwc(dy, dt), wd, du, dx2, nhwc(z, dt), act1, nhwc(gskip, dt) if enc_form else None,
                                    ops.ACT_RELU if enc_form else ops.ACT_NONE, f(scale), f(shift), f(mean.float()), f(rstd.float()),
                                    part, f(gamma), sums, dgamma, dbeta, dz)
            torch.cuda.synchronize()
            res[fused] = (from_nhwc(dz, N, H, H, C1), dgamma.cpu() - 1.0, dbeta.cpu() + 1.0, sums.cpu(),
                          from_nhwc(dx2, N, H, H, C2) if C2 else None)
        finally:
            ops.set_tunable("bn_fuse_small")
    scale_dz = float(zr.grad.abs().max())
    for fused, (dz, dg, db, sums, dx2) in res.items():
        # bf16 storage of du and dz: two roundings of 2^-9 relative each, on values up to max|dz|
        assert float((dz - zr.grad).abs().max()) <= 2.0 ** -6 * scale_dz, (name, fused, float((dz - zr.grad).abs().max()), scale_dz)
        assert torch.allclose(dg, gr.grad, rtol=2e-3, atol=2e-3 * float(gr.grad.abs().max())), (name, fused, "dgamma")
        assert torch.allclose(db, br.grad, rtol=2e-3, atol=2e-3 * float(br.grad.abs().max())), (name, fused, "dbeta")
        if C2:      # the skip half of the input gradient is a plain convolution gradient: exact on integers
            xg = torch.zeros(N, Cin, H, H, requires_grad=True)
            yy = F.conv_transpose2d(xg, w, None, stride=2, padding=1) if tr else F.conv2d(xg, w, None, stride=2, padding=1)
            yy.backward(dy)
            assert torch.equal(dx2, xg.grad[:, C1:].bfloat16().float()), (name, fused, "dx2")
    for k in range(4):      # dz, dgamma, dbeta, sums: one launch against two, bit for bit
        assert torch.equal(res[1][k], res[0][k]), (name, k)
