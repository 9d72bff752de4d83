"""GPU parity: every convolution-family entry point of the C ABI against plain PyTorch-CPU fp32
(F.conv2d / F.conv_transpose2d and their autograd), on shapes that hit each kernel:
vector-ALU tile kernel (fp32, and the tiny-Cin layers), row-dot kernels (Cout = 1),
bf16 MFMA forward / input-gradient / weight-gradient kernels.

Tolerances: fp32 storage 1e-4 relative (north_star); bf16 storage 1.5e-2 relative on tensors whose
inputs were pre-rounded to bf16 (output rounding 2^-9 plus fp32 accumulation-order noise)."""
import pytest
import torch
import torch.nn.functional as F

from _gpu_util import dev, fwd_pack, from_nhwc, max_err, nhwc, q, rel_err, rnd, unpack_fwd

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 1.5e-2}

# (name, transposed, stride, N, H, W, C1, C2, Cout, relu1, relu2)
CASES = [
    ("enc0_cin1", 0, 2, 3, 16, 16, 1, 0, 64, 0, 0),
    ("disc0_cin1x2", 0, 2, 2, 32, 32, 1, 1, 64, 0, 0),
    ("enc_mid", 0, 2, 2, 16, 16, 64, 0, 128, 0, 0),
    ("enc_wide", 0, 2, 3, 8, 8, 128, 0, 256, 0, 0),
    ("enc_bottleneck", 0, 2, 5, 2, 2, 128, 0, 128, 0, 0),
    ("dec0", 1, 2, 3, 1, 1, 128, 0, 128, 1, 0),
    ("dec_skip", 1, 2, 2, 4, 4, 128, 64, 64, 0, 1),
    ("dec_skip_wide", 1, 2, 2, 8, 8, 128, 128, 128, 0, 1),
    ("head_cout1", 1, 2, 2, 16, 16, 64, 64, 1, 0, 0),
    ("patch_final", 0, 1, 2, 6, 6, 64, 0, 1, 0, 0),
    ("patch_final_wide", 0, 1, 3, 16, 16, 256, 0, 1, 0, 0),
    ("head_relu_skip", 1, 2, 2, 8, 8, 32, 32, 1, 1, 1),
    ("odd_batch_rgb", 0, 2, 3, 8, 8, 3, 3, 64, 0, 0),
    # enough 8 x 16 output tiles for the patch-resident MFMA kernel (no split-K): stride-2 windows forward,
    # per-phase windows in the input gradient (64-wide tile), and the transposed pair with two sources
    ("enc_patch", 0, 2, 4, 128, 128, 64, 0, 256, 0, 0),
    ("dec_patch", 1, 2, 4, 64, 64, 64, 64, 128, 1, 1),
    # >= 512 tiles of 16 x 16 output pixels: the 8-wave variant of the patch kernel (forward of both)
    ("enc_patch256", 0, 2, 8, 256, 256, 64, 0, 128, 0, 0),
    ("dec_patch256", 1, 2, 8, 64, 64, 128, 0, 128, 1, 0),
    # thin layers whose wide side has 64-pixel rows: the weight gradient stages its thin patch through LDS
    ("enc0_wide", 0, 2, 2, 128, 128, 1, 0, 64, 0, 0),
    ("disc0_wide", 0, 2, 2, 128, 128, 1, 1, 64, 0, 0),
    ("head_wide", 1, 2, 2, 64, 64, 64, 64, 1, 1, 1),
    # 64 output channels on 16 x 16 pixel tiles (gg_fwd_patch_k<256, 64>): forward here, input gradient in enc_patch256
    ("dec_patch256x64", 1, 2, 8, 64, 64, 128, 128, 64, 0, 1),
    # input gradient on the 16 x 16 patch kernel (dgrad of a stride-2 conv = 4 phases x 128 tiles)
    ("enc_dgrad256", 0, 2, 8, 128, 128, 128, 0, 128, 0, 0),
    # long reduction, few output rows: split-K slabs + finish kernel in forward and input gradient
    ("enc_splitk", 0, 2, 4, 8, 8, 256, 0, 256, 0, 0),
    ("dec_splitk", 1, 2, 4, 4, 4, 256, 256, 256, 1, 1),
]


# Kernel family (pai_conv_kernel_id: 0 vector-ALU tile, 1 row-dot, 2 / 3 bf16 matrix-core tile 128- / 64-wide, 4 thin-layer
# matrix-core kernels) every bf16 case runs for forward / input gradient / weight gradient: a silent dispatch change
# must not leave this test green on another kernel.  (tests/test_abi.py checks the same table without a GPU.)
BF16_FAMILY = {
    "enc0_cin1": (4, 1, 4), "disc0_cin1x2": (4, 1, 4), "enc_mid": (2, 3, 2), "enc_wide": (2, 2, 2),
    "enc_bottleneck": (2, 2, 2), "dec0": (2, 2, 2), "dec_skip": (3, 3, 3), "dec_skip_wide": (2, 2, 2),
    "head_cout1": (1, 4, 4), "patch_final": (1, 4, 1), "patch_final_wide": (1, 4, 4), "head_relu_skip": (1, 4, 4),
    "odd_batch_rgb": (0, 0, 0), "enc_patch": (2, 3, 2), "dec_patch": (2, 3, 2), "enc_patch256": (2, 3, 2),
    "dec_patch256": (2, 2, 2), "enc0_wide": (4, 1, 4), "disc0_wide": (4, 1, 4), "head_wide": (1, 4, 4),
    "dec_patch256x64": (3, 2, 3), "enc_dgrad256": (2, 2, 2), "enc_splitk": (2, 2, 2), "dec_splitk": (2, 2, 2),
}


def _ref(case, x1, x2, w, b):
    name, tr, s, N, H, W, C1, C2, Cout, r1, r2 = case
    a1 = F.relu(x1) if r1 else x1
    xs = [a1]
    if C2:
        xs.append(F.relu(x2) if r2 else x2)
    x = torch.cat(xs, 1).requires_grad_(True)
    w = w.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True) if b is not None else None
    y = F.conv_transpose2d(x, w, b, stride=2, padding=1) if tr else F.conv2d(x, w, b, stride=s, padding=1)
    return x, w, b, y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_family(pai, case, dtype):
    from thesis_pai_reconstruction_amd import ops
    name, tr, s, N, H, W, C1, C2, Cout, r1, r2 = case
    Cin = C1 + C2
    x1 = q(rnd((N, C1, H, W), 1), dtype)
    x2 = q(rnd((N, C2, H, W), 2), dtype) if C2 else None
    wshape = (Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4)
    w = q(rnd(wshape, 3, 0.05), dtype)
    bias = rnd((Cout,), 4, 0.1)
    x, wr, br, y = _ref(case, x1, x2, w, bias)
    OH, OW = y.shape[2], y.shape[3]
    dy = q(rnd(tuple(y.shape), 5), dtype)
    y.backward(dy)

    d = ops.make_desc(dtype, tr, N, H, W, C1, C2, Cout, s, r1, r2, ops.ACT_LRELU)
    assert ops.conv_out_hw(d) == (OH, OW)
    if dtype == torch.bfloat16:
        # (the table is the dispatch WITHOUT registered scratch; with it -- earlier tests of this process registered
        # some -- the thin layers move from the row-dot / vector-ALU fallbacks to the thin matrix-core kernels)
        got = tuple(ops.conv_kernel_id(d, op) for op in (0, 1, 2))
        assert all(g == w or (w in (0, 1) and g == 4) for g, w in zip(got, BF16_FAMILY[name])), (name, got)
    # split-K scratch: small-M / long-K layers take the split path only when it is registered
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    wm = fwd_pack(w, bool(tr))
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, Cout, 16, Cin, wf, wd)
    X1 = nhwc(x1, dtype)
    X2 = nhwc(x2, dtype) if C2 else None
    B = bias.to(dev())

    # ---- forward: raw, activated, fp32 outputs + BN partial statistics -------------------------
    y_raw = torch.empty(N * OH * OW * Cout, dtype=dtype, device=dev())
    y_act = torch.empty_like(y_raw)
    want_f32 = Cout <= 2 or dtype == torch.float32
    rows = ops.conv_fwd_stats_rows(d)
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, dtype=torch.float32,
                        device=dev())
    use_stats = Cout > 2
    ops.conv_fwd(d, X1, X2, wf, B, y_raw=y_raw, stats=stats if use_stats else None)
    ops.conv_fwd(d, X1, X2, wf, B, y_act=y_act)
    torch.cuda.synchronize()
    got = from_nhwc(y_raw, N, OH, OW, Cout)
    assert rel_err(got, y.detach()) < TOL[dtype], name
    got_act = from_nhwc(y_act, N, OH, OW, Cout)
    assert rel_err(got_act, F.leaky_relu(y.detach(), 0.2)) < TOL[dtype], name
    if want_f32:
        y32 = torch.empty(N * OH * OW * Cout, dtype=torch.float32, device=dev())
        ops.conv_fwd(d, X1, X2, wf, B, y_f32=y32)
        assert rel_err(from_nhwc(y32, N, OH, OW, Cout), F.leaky_relu(y.detach(), 0.2)) < (1e-4 if dtype == torch.float32 else 5e-3)
    if use_stats:
        st = stats[:rows * 2 * Cout].view(rows, 2, Cout).double().sum(0).cpu()
        yd = y.detach().double()
        assert rel_err(st[0], yd.sum((0, 2, 3))) < 1e-4 or float((st[0] - yd.sum((0, 2, 3))).abs().max()) < 1e-2
        assert rel_err(st[1], (yd * yd).sum((0, 2, 3))) < 1e-4

    # ---- input gradient (split over the two sources) ----------------------------------------------
    DY = nhwc(dy, dtype)
    dx1 = torch.empty(N * H * W * C1, dtype=dtype, device=dev())
    dx2 = torch.empty(N * H * W * C2, dtype=dtype, device=dev()) if C2 else None
    ops.conv_dgrad(d, DY, wd, dx1, dx2)
    torch.cuda.synchronize()
    gx = x.grad  # gradient w.r.t. the (already relu'd) concatenated input
    assert rel_err(from_nhwc(dx1, N, H, W, C1), gx[:, :C1]) < TOL[dtype], name
    if C2:
        assert rel_err(from_nhwc(dx2, N, H, W, C2), gx[:, C1:]) < TOL[dtype], name
        dx2b = torch.zeros_like(dx2)
        ops.conv_dgrad(d, DY, wd, None, dx2b, only_c2=True)
        assert rel_err(from_nhwc(dx2b, N, H, W, C2), gx[:, C1:]) < TOL[dtype], name

    # fused activation backward in the dgrad store == dgrad, then pai_act_bwd (bit-identical)
    A1 = nhwc(q(rnd((N, C1, H, W), 6), dtype), dtype)
    two = torch.empty_like(dx1)
    ops.act_bwd(dtype, dx1, ops.ACT_LRELU, None, ops.ACT_NONE, A1, dx1.numel(), two)
    one = torch.zeros_like(dx1)
    dx2c = torch.zeros_like(dx2) if C2 else None
    ops.conv_dgrad_act(d, DY, wd, one, dx2c, A1, ops.ACT_LRELU)
    torch.cuda.synchronize()
    # (split-K included: the per-split slabs are summed in a fixed order)
    assert torch.equal(one, two), name
    if C2:
        assert torch.equal(dx2c, dx2), name

    # producer backward in the dgrad store (pai_conv_dgrad_bn) == dgrad, then pai_act_bwd / pai_bn_bwd_reduce
    if C1 % 8 == 0 and ((C1 // 8) & (C1 // 8 - 1)) == 0:
        M = N * H * W
        Z = nhwc(q(rnd((N, C1, H, W), 7), dtype), dtype)
        ADD = nhwc(q(rnd((N, C1, H, W), 8), dtype), dtype)
        # (a) no norm: du = lrelu'(z) * g + add
        two = torch.empty_like(dx1)
        ops.act_bwd(dtype, dx1, ops.ACT_LRELU, ADD, ops.ACT_NONE, Z, dx1.numel(), two)
        one = torch.zeros_like(dx1)
        dx2c = torch.zeros_like(dx2) if C2 else None
        rows = ops.conv_dgrad_bn(d, DY, wd, one, dx2c, Z, ops.ACT_LRELU, ADD, ops.ACT_NONE)
        torch.cuda.synchronize()
        assert rows == 0 and torch.equal(one, two), name
        if C2:
            assert torch.equal(dx2c, dx2), name
        # (b) BatchNorm producer read through LeakyReLU by this layer and through ReLU by a skip consumer
        mean = rnd((C1,), 9, 0.3).to(dev())
        rstd = (rnd((C1,), 10, 0.2).abs() + 0.5).to(dev())
        gamma = (rnd((C1,), 11, 0.5) + 1.0).to(dev())
        beta = rnd((C1,), 12, 0.3).to(dev())
        scale = gamma * rstd
        shift = beta - mean * scale
        Aact = torch.empty_like(Z)
        ops.bn_apply(dtype, Z, M, C1, scale, shift, ops.ACT_LRELU, Aact)
        du2 = torch.empty_like(dx1)
        part2 = torch.zeros(ops.bn_bwd_partial_rows(M) * 2 * C1, dtype=torch.float32, device=dev())
        sums2 = torch.zeros(2 * C1, dtype=torch.float32, device=dev())
        dg2, db2 = torch.zeros(C1, device=dev()), torch.zeros(C1, device=dev())
        ops.bn_bwd_reduce(dtype, dx1, ops.ACT_LRELU, ADD, ops.ACT_RELU, Aact, Z, M, C1, mean, rstd, du2, part2, sums2,
                          dg2, db2)
        du1 = torch.zeros_like(dx1)
        part1 = torch.zeros(ops.conv_dgrad_bn_rows_max(d) * 2 * C1, dtype=torch.float32, device=dev())
        sums1 = torch.zeros(2 * C1, dtype=torch.float32, device=dev())
        dg1, db1 = torch.zeros(C1, device=dev()), torch.zeros(C1, device=dev())
        rows = ops.conv_dgrad_bn(d, DY, wd, du1, dx2c, Z, ops.ACT_LRELU, ADD, ops.ACT_RELU, scale, shift, mean, rstd,
                                 part1)
        assert 0 < rows <= ops.conv_dgrad_bn_rows_max(d)
        ops.bn_bwd_finalize(part1, rows, C1, sums1, dg1, db1)
        torch.cuda.synchronize()
        assert torch.equal(du1, du2), name
        if C2:
            assert torch.equal(dx2c, dx2), name
        scl = float(sums2.abs().max()) + 1e-6
        assert float((sums1 - sums2).abs().max()) / scl < 1e-4, name
        assert torch.allclose(dg1, sums1[C1:]) and torch.allclose(db1, sums1[:C1])

    # ---- weight / bias gradient -----------------------------------------------------------------------
    dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
    db = torch.zeros(Cout, dtype=torch.float32, device=dev())
    ops.conv_wgrad(d, X1, X2, DY, dw, db)
    torch.cuda.synchronize()
    tol_w = 1e-4 if dtype == torch.float32 else 3e-3  # fp32 accumulate of exact bf16 products
    assert rel_err(unpack_fwd(dw, Cout, Cin, bool(tr)), wr.grad) < tol_w, name
    assert rel_err(db.cpu(), br.grad) < tol_w, name
    # accumulation semantics: a second call adds
    ops.conv_wgrad(d, X1, X2, DY, dw, None)
    torch.cuda.synchronize()
    assert rel_err(unpack_fwd(dw, Cout, Cin, bool(tr)), 2 * wr.grad) < tol_w, name


# (name, N, H, W, Cin, Cout): vector-ALU kernel (1 -> 64), MFMA forward / input gradient with the vector-ALU weight
# gradient (9 * 64 is not a multiple of 128), MFMA everywhere (128 -> 128), row-dot forward (64 -> 1)
CONV3 = [("in_conv", 2, 16, 16, 1, 64), ("c64", 2, 16, 24, 64, 64), ("c128", 3, 16, 16, 128, 128),
         ("out_conv", 2, 16, 16, 64, 1), ("in_conv_ragged", 3, 10, 12, 1, 64), ("out_conv_ragged", 1, 6, 20, 64, 1),
         ("in_conv_big", 2, 64, 64, 1, 64), ("out_conv_big", 2, 64, 64, 64, 1),
         # 9 x Cin not a multiple of the 128-column weight-gradient tile: partly empty last column block
         ("c64_pow2", 2, 16, 16, 64, 64), ("c192_64", 2, 8, 8, 192, 64), ("c64_128", 1, 32, 32, 64, 128)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV3, ids=[c[0] for c in CONV3])
def test_conv3x3(pai, case, dtype):
    """nn.Conv2d(kernel_size=3, padding=1) of the residual U-Net (reference models/res_unet.py:59,62,265,308)
    through the same gather-GEMM entry points (kernel = 3: 9 taps, stride 1)."""
    name, N, H, W, C, K = case
    _same_conv_case(3, N, H, W, C, K, dtype)


# (name, kernel, N, H, W, Cin, Cout): the 16- / 32-channel bottleneck convolutions of the TransUNet / ResNet-50 encoder
# blocks (reference models/trans_unet.py:203-227, models/res_unet.py:86-95) -- in bf16 the no-LDS MFMA kernel of
# gg_small.hip (forward, input gradient) and the partly filled tiles of gg_wgrad_mfma_k; ragged pixel counts and
# non-power-of-two images included (the latter keep the vector-ALU weight gradient)
SMALL = [("pw64_16", 1, 2, 16, 16, 64, 16), ("pw16_64", 1, 2, 16, 16, 16, 64), ("pw16_128", 1, 1, 8, 32, 16, 128),
         ("pw128_32", 1, 2, 8, 8, 128, 32), ("pw32_256", 1, 3, 4, 4, 32, 256), ("pw256_32", 1, 2, 4, 8, 256, 32),
         ("c16", 3, 2, 16, 16, 16, 16), ("c32", 3, 3, 8, 16, 32, 32), ("c16_ragged", 3, 1, 6, 10, 16, 16),
         ("pw64_48", 1, 1, 5, 7, 64, 48), ("c32_16", 3, 2, 8, 8, 32, 16)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_small_channel_convs(pai, case, dtype):
    from thesis_pai_reconstruction_amd import ops
    name, k, N, H, W, C, K = case
    d = ops.make_desc(dtype, 0, N, H, W, C, 0, K, 1, 0, 0, ops.ACT_NONE, kernel=k)
    if dtype == torch.bfloat16:
        assert ops.conv_kernel_id(d, 0) == 5 and ops.conv_kernel_id(d, 1) == 5, name    # small_mfma_bf16
    _same_conv_case(k, N, H, W, C, K, dtype)


def _same_conv_case(k, N, H, W, C, K, dtype):
    from thesis_pai_reconstruction_amd import ops
    tol = TOL[dtype]
    x = q(rnd((N, C, H, W), 1), dtype).requires_grad_(True)
    w = q(rnd((K, C, k, k), 2, 0.05), dtype).requires_grad_(True)
    b = rnd((K,), 3, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=k // 2)
    dy = q(rnd(tuple(y.shape), 4), dtype)
    y.backward(dy)
    d = ops.make_desc(dtype, 0, N, H, W, C, 0, K, 1, 0, 0, ops.ACT_RELU, kernel=k)
    assert ops.conv_out_hw(d) == (H, W)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    if dtype == torch.bfloat16 and k == 3 and C == 1:      # in_conv: thin -> wide kernels of gg_thin.hip (9 taps)
        assert ops.conv_kernel_id(d, 0) == 4 and ops.conv_kernel_id(d, 2) == 4
    if dtype == torch.bfloat16 and k == 3 and K == 1 and C % 32 == 0:     # out conv: skinny GEMM + gather, thin dgrad / wgrad
        assert [ops.conv_kernel_id(d, op) for op in (0, 1, 2)] == [4, 4, 4]
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev())          # fwd pack [K][k][k][C]
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, K, k * k, C, wf, wd)
    X, DY = nhwc(x.detach(), dtype), nhwc(dy, dtype)
    if K > 2:
        y_raw = torch.empty(N * H * W * K, dtype=dtype, device=dev())
        y_act = torch.empty_like(y_raw)
        rows = ops.conv_fwd_stats_rows(d)
        stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * K, dtype=torch.float32,
                            device=dev())
        ops.conv_fwd(d, X, None, wf, b.detach().to(dev()), y_raw=y_raw, stats=stats)
        ops.conv_fwd(d, X, None, wf, b.detach().to(dev()), y_act=y_act)
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(y_raw, N, H, W, K), y.detach()) < tol
        assert rel_err(from_nhwc(y_act, N, H, W, K), F.relu(y.detach())) < tol
        st = stats[:rows * 2 * K].view(rows, 2, K).double().sum(0).cpu()
        assert rel_err(st[1], (y.detach().double() ** 2).sum((0, 2, 3))) < 1e-4
    else:
        y32 = torch.empty(N * H * W * K, dtype=torch.float32, device=dev())
        d.epilogue_act = ops.ACT_TANH
        ops.conv_fwd(d, X, None, wf, b.detach().to(dev()), y_f32=y32)
        assert rel_err(from_nhwc(y32, N, H, W, K), torch.tanh(y.detach())) < (1e-4 if dtype == torch.float32 else 5e-3)
    dx = torch.empty(N * H * W * C, dtype=dtype, device=dev())
    ops.conv_dgrad(d, DY, wd, dx, None)
    assert rel_err(from_nhwc(dx, N, H, W, C), x.grad) < tol
    dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
    db = torch.zeros(K, dtype=torch.float32, device=dev())
    ops.conv_wgrad(d, X, None, DY, dw, db)
    torch.cuda.synchronize()
    tol_w = 1e-4 if dtype == torch.float32 else 3e-3
    assert rel_err(dw.cpu().view(K, k, k, C).permute(0, 3, 1, 2), w.grad) < tol_w
    assert rel_err(db.cpu(), b.grad) < tol_w
    # overwrite form: same values into buffers that hold garbage
    dw2 = torch.full_like(dw, float("nan"))
    db2 = torch.full_like(db, 1e30)
    ops.conv_wgrad_overwrite(d, X, None, DY, dw2, db2)
    torch.cuda.synchronize()
    assert rel_err(dw2.cpu(), dw.cpu()) < 1e-6 and rel_err(db2.cpu(), db.cpu()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("C,groups,N,H,W", [(128, 32, 2, 16, 16), (128, 32, 3, 24, 48), (32, 8, 3, 8, 16), (128, 32, 1, 6, 10)])
def test_grouped_conv3x3(pai, dtype, C, groups, N, H, W):
    """nn.Conv2d(C, C, 3, padding=1, groups=C/4) of ResidualBlockNeXt (reference models/res_unet.py:151-157): block-diagonal
    dense packs + the descriptor's ``groups`` hint.  In bf16, 128 channels on 8 x 16-tileable images run forward and input
    gradient on grouped3_k (16-channel slices, patch in LDS); every other case takes the dense kernels.  The result must
    equal F.conv2d(groups=...) autograd either way."""
    from thesis_pai_reconstruction_amd import nnops, ops
    tol = TOL[dtype]
    x = q(rnd((N, C, H, W), 1), dtype).requires_grad_(True)
    w = q(rnd((C, C // groups, 3, 3), 2, 0.1), dtype).requires_grad_(True)
    b = rnd((C,), 3, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1, groups=groups)
    dy = q(rnd(tuple(y.shape), 4), dtype)
    y.backward(dy)
    d = ops.make_desc(dtype, 0, N, H, W, C, 0, C, 1, 0, 0, ops.ACT_RELU, kernel=3, groups=groups)
    if dtype == torch.bfloat16 and C == 128 and H % 8 == 0 and W % 16 == 0:
        assert ops.conv_kernel_id(d, 0) == 6 and ops.conv_kernel_id(d, 1) == 6
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    wm = nnops._dense_fwd_pack(w.detach().to(dev()), groups)                 # [C][3][3][C], block-diagonal
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, C, 9, C, wf, wd)
    X, DY = nhwc(x.detach(), dtype), nhwc(dy, dtype)
    y_raw = torch.empty(N * H * W * C, dtype=dtype, device=dev())
    y_act = torch.empty_like(y_raw)
    rows = ops.conv_fwd_stats_rows(d)
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * C, dtype=torch.float32, device=dev())
    ops.conv_fwd(d, X, None, wf, b.detach().to(dev()), y_raw=y_raw, stats=stats)
    ops.conv_fwd(d, X, None, wf, b.detach().to(dev()), y_act=y_act)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y_raw, N, H, W, C), y.detach()) < tol
    assert rel_err(from_nhwc(y_act, N, H, W, C), F.relu(y.detach())) < tol
    st = stats[:rows * 2 * C].view(rows, 2, C).double().sum(0).cpu()
    assert rel_err(st[0], y.detach().double().sum((0, 2, 3))) < 1e-3
    assert rel_err(st[1], (y.detach().double() ** 2).sum((0, 2, 3))) < 1e-4
    dx = torch.empty(N * H * W * C, dtype=dtype, device=dev())
    ops.conv_dgrad(d, DY, wd, dx, None)
    assert rel_err(from_nhwc(dx, N, H, W, C), x.grad) < tol
    dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
    ops.conv_wgrad(d, X, None, DY, dw, None)
    torch.cuda.synchronize()
    gw = nnops._grad_from_fwd_pack(dw, w, groups)                           # block-diagonal blocks -> [C][C/g][3][3]
    assert rel_err(gw.cpu(), w.grad) < (1e-4 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("N,H,W", [(2, 8, 32), (3, 12, 64), (9, 128, 64)])
def test_grouped_conv3x3_weight_gradient_exact(pai, N, H, W):
    """Forward (with bias and BatchNorm partial statistics) and input gradient on grouped3_k, and the weight gradient of the
    grouped 3 x 3 convolution on grouped3_wgrad_k (diagonal 16-channel blocks only, partial
    blocks of the persistent workgroups summed in a fixed order): on small-integer data every sum is exact in fp32, so the
    group blocks must equal PyTorch-CPU's F.conv2d(groups=32) weight gradient BIT FOR BIT -- accumulating into a non-zero
    dW and overwriting a dirty one; (9, 128, 64): 576 tiles on 512 workgroups, some take two.  Without the weight-gradient
    workspace the dense kernel still serves the call."""
    from thesis_pai_reconstruction_amd import nnops, ops
    C, groups, dtype = 128, 32, torch.bfloat16
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randint(-2, 3, (N, C, H, W), generator=g).float().requires_grad_(True)
    w = torch.randint(-2, 3, (C, C // groups, 3, 3), generator=g).float().requires_grad_(True)
    dy = torch.randint(-2, 3, (N, C, H, W), generator=g).float()
    F.conv2d(x, w, None, padding=1, groups=groups).backward(dy)
    d = ops.make_desc(dtype, 0, N, H, W, C, 0, C, 1, 0, 0, ops.ACT_NONE, kernel=3, groups=groups)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    ops.ensure_wgrad_workspace([d], dev())
    assert ops.conv_wgrad_workspace_bytes(d) > 0
    assert ops.conv_kernel_id(d, 2) == 6 and ops.conv_kernel_name(d, 2) == "grouped3_wgrad_k"
    X, DY = nhwc(x.detach(), dtype), nhwc(dy, dtype)
    # forward and input gradient on grouped3_k (wave = slice, filter fragments resident), bit for bit as well
    if H % 8 == 0 and W % 16 == 0:
        assert ops.conv_kernel_id(d, 0) == 6 and ops.conv_kernel_id(d, 1) == 6
    wm = nnops._dense_fwd_pack(w.detach().to(dev()), groups)
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, C, 9, C, wf, wd)
    bias = torch.randint(-8, 9, (C,), generator=g).float()
    yo = torch.full((N * H * W * C,), 7.0, dtype=dtype, device=dev())
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * C, dtype=torch.float32, device=dev())
    ops.conv_fwd(d, X, None, wf, bias.to(dev()), y_raw=yo, stats=stats)
    dxo = torch.full((N * H * W * C,), 7.0, dtype=dtype, device=dev())
    ops.conv_dgrad(d, DY, wd, dxo, None)
    torch.cuda.synchronize()
    y_ref = F.conv2d(x.detach(), w.detach(), bias, padding=1, groups=groups)
    assert torch.equal(from_nhwc(yo, N, H, W, C), y_ref.to(dtype).float())
    rows = ops.conv_fwd_stats_rows(d)
    st = stats[:rows * 2 * C].view(rows, 2, C).double().sum(0).cpu()
    assert torch.equal(st[0], y_ref.double().sum((0, 2, 3))) and torch.equal(st[1], (y_ref.double() ** 2).sum((0, 2, 3)))
    assert torch.equal(from_nhwc(dxo, N, H, W, C), x.grad.to(dtype).float())
    base = torch.randint(-3, 4, (C * 9 * C,), generator=g).float().to(dev())
    dw = base.clone()
    ops.conv_wgrad(d, X, None, DY, dw, None)                    # dW += ...
    got = nnops._grad_from_fwd_pack(dw, w, groups) - nnops._grad_from_fwd_pack(base, w, groups)
    assert torch.equal(got.cpu(), w.grad)
    dw2 = torch.full_like(base, 12345.0)
    ops.conv_wgrad_overwrite(d, X, None, DY, dw2, None)         # dW = ...
    assert torch.equal(nnops._grad_from_fwd_pack(dw2, w, groups).cpu(), w.grad)
    # the dense kernel (tunable off) agrees on the group blocks
    try:
        ops.set_tunable("grouped_wgrad", 0)
        assert ops.conv_kernel_id(d, 2) != 6
        dw3 = torch.zeros_like(base)
        ops.conv_wgrad(d, X, None, DY, dw3, None)
        assert torch.equal(nnops._grad_from_fwd_pack(dw3, w, groups).cpu(), w.grad)
    finally:
        ops.set_tunable("grouped_wgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("k,N,H,W,C1,C2,K", [(1, 2, 16, 16, 128, 128, 128), (3, 2, 16, 16, 64, 64, 64), (3, 1, 6, 10, 128, 64, 64),
                                             (1, 3, 8, 8, 256, 256, 64), (3, 2, 8, 16, 64, 128, 128)])
def test_same_conv_two_sources(pai, dtype, k, N, H, W, C1, C2, K):
    """1x1 / 3x3 "same" convolutions over two tensors read as one concatenation (the torch.cat in front of the decoder
    blocks of the residual / Trans U-Nets, reference models/res_unet.py:327, models/trans_unet.py:113): forward, both input
    gradients and the weight gradient against F.conv2d on the concatenated tensor."""
    from thesis_pai_reconstruction_amd import ops
    tol = TOL[dtype]
    x1 = q(rnd((N, C1, H, W), 1), dtype).requires_grad_(True)
    x2 = q(rnd((N, C2, H, W), 5), dtype).requires_grad_(True)
    w = q(rnd((K, C1 + C2, k, k), 2, 0.05), dtype).requires_grad_(True)
    b = rnd((K,), 3, 0.1).requires_grad_(True)
    y = F.conv2d(torch.cat([x1, x2], 1), w, b, padding=k // 2)
    dy = q(rnd(tuple(y.shape), 4), dtype)
    y.backward(dy)
    d = ops.make_desc(dtype, 0, N, H, W, C1, C2, K, 1, 0, 0, ops.ACT_NONE, kernel=k)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, K, k * k, C1 + C2, wf, wd)
    X1, X2, DY = nhwc(x1.detach(), dtype), nhwc(x2.detach(), dtype), nhwc(dy, dtype)
    y_raw = torch.empty(N * H * W * K, dtype=dtype, device=dev())
    rows = ops.conv_fwd_stats_rows(d)
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * K, dtype=torch.float32, device=dev())
    ops.conv_fwd(d, X1, X2, wf, b.detach().to(dev()), y_raw=y_raw, stats=stats)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(y_raw, N, H, W, K), y.detach()) < tol
    st = stats[:rows * 2 * K].view(rows, 2, K).double().sum(0).cpu()
    assert rel_err(st[1], (y.detach().double() ** 2).sum((0, 2, 3))) < 1e-4
    dx1 = torch.empty(N * H * W * C1, dtype=dtype, device=dev())
    dx2 = torch.empty(N * H * W * C2, dtype=dtype, device=dev())
    ops.conv_dgrad(d, DY, wd, dx1, dx2)
    assert rel_err(from_nhwc(dx1, N, H, W, C1), x1.grad) < tol
    assert rel_err(from_nhwc(dx2, N, H, W, C2), x2.grad) < tol
    dw = torch.full((wm.numel(),), float("nan"), dtype=torch.float32, device=dev())
    db = torch.zeros(K, dtype=torch.float32, device=dev())
    ops.conv_wgrad_overwrite(d, X1, X2, DY, dw, db)
    torch.cuda.synchronize()
    tol_w = 1e-4 if dtype == torch.float32 else 3e-3
    assert rel_err(dw.cpu().view(K, k, k, C1 + C2).permute(0, 3, 1, 2), w.grad) < tol_w
    assert rel_err(db.cpu(), b.grad) < tol_w


def test_bad_arguments_fail_loudly(pai):
    from thesis_pai_reconstruction_amd import ops
    d = ops.make_desc(torch.float32, 0, 1, 7, 8, 1, 0, 64, 2)
    with pytest.raises(ops.PaiError):
        ops.conv_out_hw(d)          # odd height for a stride-2 conv
    d = ops.make_desc(torch.float32, 0, 1, 8, 8, 1, 0, 64, 2)
    with pytest.raises(ops.PaiError):
        ops.conv_fwd(d, torch.zeros(64), None, torch.zeros(64), None, y_raw=torch.zeros(64))  # CPU tensors
