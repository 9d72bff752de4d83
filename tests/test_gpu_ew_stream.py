"""GPU: the streaming forms of the big bf16 tensor passes (csrc/ew_stream.hip) against the generic kernels they replace and
against fp64 formulas -- nn.BatchNorm2d forward / backward around ReLU, the residual-block tail and the residual sum of the
reference's blocks (models/res_unet.py:133-171, models/pix2pix.py:70,106; aten::native_batch_norm_backward).

``pai_set_tunable("ew_stream", 0)`` sends a call to the generic kernel: forward-side passes must agree BIT FOR BIT (same
operations in the same order), the BatchNorm backward (re-associated: dz = A du + (B (z - mean) + K)) to one bf16 rounding
of the generic kernel's result (plus the fp32 cancellation error where dz is tiny) and to bf16 rounding of the fp64 formula; the partial sums to fp32 summation error."""
import pytest
import torch

from _gpu_util import dev

pytestmark = pytest.mark.gpu

# rows x channels: more than 4096 rows, ragged against every stride of the kernels (256 x 4 vectors per block sweep)
SHAPES = [(4096 + 4099, 64), (40000 + 3, 128), (9001, 512), (5000, 2048), (16 * 64 * 64, 8)]
ACTS = ["none", "relu", "lrelu"]


def _act(ops, name):
    return {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}[name]


def _data(M, C, seed):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(dev())
    gr = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev())
    f32 = lambda t: t.float().to(dev()).contiguous()
    mean = f32(torch.randn(C, generator=g) * 0.2 + 0.3)
    rstd = f32(1.0 / (1.4 + 0.2 * torch.rand(C, generator=g)))
    gamma = f32(0.5 + torch.rand(C, generator=g))
    beta = f32(torch.randn(C, generator=g) * 0.1)
    scale = gamma * rstd
    shift = beta - mean * scale
    return z, gr, mean, rstd, gamma, scale, shift


@pytest.fixture
def both(pai):
    """run(fn) -> (streaming result, generic result); nt=True also forces the non-temporal instantiation"""
    from thesis_pai_reconstruction_amd import ops

    def run(fn, nt=False):
        try:
            ops.set_tunable("ew_stream", 1)
            if nt:
                ops.set_tunable("ew_stream_nt_mb", 0)
            a = fn()
            ops.set_tunable("ew_stream", 0)
            b = fn()
        finally:
            ops.set_tunable("ew_stream")
            ops.set_tunable("ew_stream_nt_mb")
        torch.cuda.synchronize()
        return a, b
    return ops, run


@pytest.mark.parametrize("nt", [False, True], ids=["plain", "nt"])
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("shape", SHAPES, ids=[f"{m}x{c}" for m, c in SHAPES])
def test_bn_apply_bit_identical(both, shape, act, nt):
    ops, run = both
    M, C = shape
    z, _, _, _, _, scale, shift = _data(M, C, 1)

    def fn():
        out = torch.full_like(z, 7.0)
        ops.bn_apply(torch.bfloat16, z, M, C, scale, shift, _act(ops, act), out)
        return out
    a, b = run(fn, nt)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    ref = z.float() * scale + shift
    ref = {"none": ref, "relu": ref.clamp_min(0), "lrelu": torch.where(ref > 0, ref, 0.2 * ref)}[act]
    assert torch.allclose(a.float(), ref, rtol=2 ** -7, atol=1e-6)


@pytest.mark.parametrize("affb", [False, True], ids=["identity_skip", "bn_skip"])
@pytest.mark.parametrize("acts", [("relu", "none"), ("none", "relu"), ("relu", "relu"), ("none", "none")], ids=str)
@pytest.mark.parametrize("shape", SHAPES[:3], ids=[f"{m}x{c}" for m, c in SHAPES[:3]])
def test_block_tail_bit_identical(both, shape, acts, affb):
    ops, run = both
    M, C = shape
    za, zb, _, _, _, sa, ha = _data(M, C, 2)
    _, _, _, _, _, sb, hb = _data(M, C, 3)

    def fn():
        out = torch.empty_like(za)
        ops.bn2_add_act(torch.bfloat16, za, sa, ha, zb, sb if affb else None, hb if affb else None, M, C, _act(ops, acts[0]),
                        _act(ops, acts[1]), out)
        return out
    for nt in (False, True):
        a, b = run(fn, nt)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("act", ACTS)
def test_residual_sum_bit_identical(both, act):
    ops, run = both
    M, C = 70001, 64
    a0, b0, *_ = _data(M, C, 4)

    def fn():
        out = torch.empty_like(a0)
        ops.add_act(torch.bfloat16, a0, b0, _act(ops, act), out)
        return out
    for nt in (False, True):
        a, b = run(fn, nt)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def _du64(gr, z, scale, shift, act):
    pre = z.double() * scale.double() + shift.double()
    g = gr.double()
    if act == "relu":
        return torch.where(pre > 0, g, torch.zeros_like(g))
    if act == "lrelu":
        return torch.where(pre > 0, g, (0.2 * gr.float()).to(torch.bfloat16).double())
    return g


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("shape", SHAPES, ids=[f"{m}x{c}" for m, c in SHAPES])
def test_bn_backward_two_passes(both, shape, act):
    """pai_bn_bwd_reduce(_affine) + pai_bn_bwd_apply(_affine) as the composable networks call them (du not stored)."""
    ops, run = both
    M, C = shape
    dt = torch.bfloat16
    z, gr, mean, rstd, gamma, scale, shift = _data(M, C, 5)
    f32 = dict(dtype=torch.float32, device=dev())

    def fn():
        part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C, **f32)
        sums = torch.empty(2 * C, **f32)
        dz = torch.empty_like(z)
        if act == "none":
            ops.bn_bwd_reduce(dt, gr, ops.ACT_NONE, None, ops.ACT_NONE, None, z, M, C, mean, rstd, None, part, sums, None, None)
            ops.bn_bwd_apply(dt, gr, z, M, C, mean, rstd, gamma, sums, dz)
        else:
            ops.bn_bwd_reduce_affine(dt, gr, _act(ops, act), None, ops.ACT_NONE, z, M, C, scale, shift, mean, rstd, None, part,
                                     sums, None, None)
            ops.bn_bwd_apply_affine(dt, gr, _act(ops, act), z, M, C, scale, shift, mean, rstd, gamma, sums, dz)
        return sums.clone(), dz
    for nt in (False, True):
        (sa, dza), (sb, dzb) = run(fn, nt)
        du = _du64(gr, z, scale, shift, act)
        xh = (z.double() - mean.double()) * rstd.double()
        s1, s2 = du.sum(0), (du * xh).sum(0)
        mag1, mag2 = du.abs().sum(0), (du * xh).abs().sum(0)
        for got in (sa, sb):
            assert float(((got[:C].double() - s1).abs() / mag1).max()) < 2e-6
            assert float(((got[C:].double() - s2).abs() / mag2).max()) < 2e-6
        ref = gamma.double() * rstd.double() * (du - s1 / M - xh * s2 / M)
        tol = 2 ** -8 * ref.abs() + 2e-6 * (du.abs() + 1)          # half a bf16 ulp + the fp32 formation of the terms
        assert bool(((dza.double() - ref).abs() <= tol).all())
        assert bool(((dzb.double() - ref).abs() <= tol).all())
        # the two kernels against each other: at most one bf16 rounding step apart, and that rarely
        fa, fb = dza.float(), dzb.float()
        assert bool(((fa - fb).abs() <= 2 ** -7 * torch.maximum(fa.abs(), fb.abs()) + 4e-6 * (du.abs().float() + 1)).all())
        assert float((fa != fb).float().mean()) < 0.05


def test_small_and_odd_layers_stay_on_the_generic_kernels(both):
    """<= 4096 rows (the one-launch forms' arithmetic) and channel counts that are not 8 x 2^k: same bits with the switch on
    and off, i.e. the same kernel ran."""
    ops, run = both
    dt = torch.bfloat16
    for M, C in ((4096, 512), (20000, 24), (20000, 192)):
        z, gr, mean, rstd, gamma, scale, shift = _data(M, C, 6)
        f32 = dict(dtype=torch.float32, device=dev())

        def fn():
            sums = torch.linspace(-1, 1, 2 * C, **f32)
            dz = torch.empty_like(z)
            ops.bn_bwd_apply_affine(dt, gr, ops.ACT_RELU, z, M, C, scale, shift, mean, rstd, gamma, sums, dz)
            return dz
        a, b = run(fn)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("act", ["relu", "none"])
@pytest.mark.parametrize("shape", [(40000 + 3, 128), (9001, 64), (4096, 64), (20000, 24)], ids=str)
def test_block_tail_backward_both_branches_in_one_pass(both, shape, act):
    """pai_bn2_bwd_reduce / pai_bn2_bwd_apply (both BatchNorms of a residual block's tail read the same gradient) against the
    two one-branch call pairs they stand for: sums to fp32 summation error, dz to bf16 rounding.  Small / odd shapes run the
    one-branch kernels inside the entry point: bit-identical."""
    ops, run = both
    M, C = shape
    dt = torch.bfloat16
    za, d, mean_a, rstd_a, gamma_a, scale_a, shift_a = _data(M, C, 7)
    zb, _, mean_b, rstd_b, gamma_b, _, _ = _data(M, C, 8)
    f32 = dict(dtype=torch.float32, device=dev())
    a = _act(ops, act)
    rows = ops.bn_bwd_partial_rows(M)
    if C / 8 not in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        pytest.skip("pai_bn_bwd_reduce needs C = 8 * 2^k")

    def pair():
        pa, pb = torch.empty(rows * 2 * C, **f32), torch.empty(rows * 2 * C, **f32)
        sa, sb = torch.empty(2 * C, **f32), torch.empty(2 * C, **f32)
        dza, dzb = torch.empty_like(za), torch.empty_like(zb)
        ops.bn2_bwd_reduce(dt, d, a, za, zb, M, C, scale_a, shift_a, mean_a, rstd_a, mean_b, rstd_b, pa, pb, sa, sb)
        ops.bn2_bwd_apply(dt, d, a, za, zb, M, C, scale_a, shift_a, mean_a, rstd_a, gamma_a, sa, mean_b, rstd_b, gamma_b, sb, dza, dzb)
        return sa.clone(), sb.clone(), dza, dzb

    def single():
        pa, pb = torch.empty(rows * 2 * C, **f32), torch.empty(rows * 2 * C, **f32)
        sa, sb = torch.empty(2 * C, **f32), torch.empty(2 * C, **f32)
        dza, dzb = torch.empty_like(za), torch.empty_like(zb)
        if act == "none":
            ops.bn_bwd_reduce(dt, d, ops.ACT_NONE, None, ops.ACT_NONE, None, za, M, C, mean_a, rstd_a, None, pa, sa, None, None)
            ops.bn_bwd_apply(dt, d, za, M, C, mean_a, rstd_a, gamma_a, sa, dza)
        else:
            ops.bn_bwd_reduce_affine(dt, d, a, None, ops.ACT_NONE, za, M, C, scale_a, shift_a, mean_a, rstd_a, None, pa, sa, None, None)
            ops.bn_bwd_apply_affine(dt, d, a, za, M, C, scale_a, shift_a, mean_a, rstd_a, gamma_a, sa, dza)
        ops.bn_bwd_reduce(dt, d, ops.ACT_NONE, None, ops.ACT_NONE, None, zb, M, C, mean_b, rstd_b, None, pb, sb, None, None)
        ops.bn_bwd_apply(dt, d, zb, M, C, mean_b, rstd_b, gamma_b, sb, dzb)
        return sa.clone(), sb.clone(), dza, dzb

    got = pair()
    want = single()
    torch.cuda.synchronize()
    big = M > 4096
    for g_, w_ in zip(got[:2], want[:2]):
        if big:
            assert float((g_ - w_).abs().max()) <= 2e-5 * float(w_.abs().max()) + 1e-3
        else:
            assert torch.equal(g_, w_)
    for g_, w_ in zip(got[2:], want[2:]):
        if big:
            fa, fb = g_.float(), w_.float()
            assert bool(((fa - fb).abs() <= 2 ** -7 * torch.maximum(fa.abs(), fb.abs()) + 4e-6 * (d.abs().float() + 1)).all())
        else:
            assert torch.equal(g_.view(torch.int16), w_.view(torch.int16))
