"""GPU parity of the BENCHMARKED bf16 matrix-core kernels, bit for bit.

The bf16 MFMA kernels that produce the bench number cannot be held to 1e-4 on random data (bf16 output rounding alone
is 2^-9).  On small-INTEGER data they can be held to ZERO: every product and every partial sum is an integer below
2^24, hence exact in fp32 whatever the summation order, tile shape or K split -- so the bf16 outputs of
pai_conv_fwd / pai_conv_dgrad must equal bf16(F.conv2d(...)) of PyTorch-CPU fp32 bit for bit, and the fp32 weight
gradients must be equal exactly.  That pins patch geometry, tap tables, phase decomposition, two-pointer (concat-free)
inputs / outputs, ReLU-on-load, split-K slabs and the fp32 atomic / read-modify-write accumulation of the weight
gradient of exactly the kernels named in each case -- including the BASELINE configs[1] layer shapes at full batch
(decoders[4-6], discriminator blocks 1-3, encoders[2, 4]) -- far tighter than the 1e-4 of the fp32 parity mode.

Reference call sites: nn.Conv2d / nn.ConvTranspose2d k4 s2 p1 of models/pix2pix.py:58-111, models/wrapper.py:229-232.
"""
import pytest
import torch
import torch.nn.functional as F

from _gpu_util import dev, fwd_pack, from_nhwc, nhwc, unpack_fwd

pytestmark = pytest.mark.gpu

SPLIT = ("gg_fwd_mfma_k<128, 128, true, true, 64>", "gg_fwd_mfma_k<128, 128, false, false, 64>")
# (name, transposed, N, H, C1, C2, Cout, relu, (forward, input-gradient, weight-gradient kernel))
CASES = [
    ("enc_patch", 0, 4, 128, 64, 0, 256, 0, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("dec_patch", 1, 4, 64, 64, 64, 128, 1, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("enc_patch256", 0, 8, 256, 64, 0, 128, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("dec_patch256", 1, 8, 64, 128, 0, 128, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("dec_patch256x64", 1, 8, 64, 128, 128, 64, 1, ("gg_fwd_patch_k<128, 64, false>", "gg_fwd_patch_k<128, 128, true>", "gg_wgrad_patch3_k<64, 128, 16>")),
    ("enc_dgrad256", 0, 8, 128, 128, 0, 128, 0, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("enc_splitk", 0, 4, 8, 256, 0, 256, 0, (SPLIT, SPLIT, "gg_wgrad_mfma_k<64>")),      # 64-channel tiles: <= 512 un-split 128-channel tiles (wgrad_narrow, round 6)
    ("dec_splitk", 1, 4, 4, 256, 256, 256, 1, (SPLIT, SPLIT, "gg_wgrad_mfma_k<64>")),
    # BASELINE configs[1] layer shapes at the benchmark batch (64; the discriminator sees 2 x 64 in its own phase)
    ("cfg2_enc2", 0, 64, 64, 128, 0, 256, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("cfg2_enc4", 0, 64, 16, 512, 0, 512, 0, (SPLIT, SPLIT, "gg_wgrad_patch3_k<128, 64, 8>")),
    ("cfg2_dec3", 1, 64, 8, 512, 512, 512, 1, (SPLIT, SPLIT, "gg_wgrad_patch3_k<128, 64, 8>")),
    ("cfg2_dec4", 1, 64, 16, 512, 512, 256, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("cfg2_dec5", 1, 64, 32, 256, 256, 128, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("cfg2_dec6", 1, 64, 64, 128, 128, 64, 1, ("gg_fwd_patch1_k<256, 64, false>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<64, 128, 16>")),
    ("cfg2_D1", 0, 128, 128, 64, 0, 128, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("cfg2_D2", 0, 128, 64, 128, 0, 256, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
    ("cfg2_D3", 0, 128, 32, 256, 0, 512, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch3_k<128, 64, 16>")),
]


def _ints(shape, seed, lo=-2, hi=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _named(ops, d, op, want):
    got = ops.conv_kernel_name(d, op)
    assert got in ((want,) if isinstance(want, str) else want), (op, got, want)
    return got


def _reference(tr, x1, x2, w, dy, relu, bias=None):
    """PyTorch-CPU fp32 (exact on this data): y, dx (w.r.t. the relu'd, concatenated input), dw."""
    xs = [F.relu(x1) if relu else x1]
    if x2 is not None:
        xs.append(F.relu(x2) if relu else x2)
    x = torch.cat(xs, 1).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv_transpose2d(x, wr, bias, stride=2, padding=1) if tr else F.conv2d(x, wr, bias, stride=2, padding=1)
    y.backward(dy)
    return y.detach(), x.grad, wr.grad


def _run_case(pai, case, tunables=(), workspace=True):
    """workspace=False: a handle WITHOUT split-K workspace, the setting the kernel names of CASES were recorded in
    (pai_conv_kernel_name on the host); True: the default handle with the workspace registered, where the library may
    pick the split-K kernels for the smaller cases -- same bits either way."""
    from thesis_pai_reconstruction_amd import ops
    name, tr, N, H, C1, C2, Cout, relu, names = case
    Cin, dt = C1 + C2, torch.bfloat16
    x1 = _ints((N, C1, H, H), 1)
    x2 = _ints((N, C2, H, H), 2) if C2 else None
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3)
    OH = H * 2 if tr else H // 2
    dy = _ints((N, Cout, OH, OH), 5)
    bias = _ints((Cout,), 7, -3, 3)           # the bias add of the epilogue is part of what is pinned
    y_ref, dx_ref, dw_ref = _reference(tr, x1, x2, w, dy, relu, bias)
    assert float(y_ref.abs().max()) < 2 ** 24 and float(dw_ref.abs().max()) < 2 ** 24     # exact in fp32

    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, relu, relu if C2 else 0, ops.ACT_NONE)
    default = ops.handle_for(dev())
    bare = None
    if workspace:
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
        ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
        ops.ensure_wgrad_workspace([d], dev())      # pixel splits of the weight gradient: slabs + ordered sum, no atomics
    else:
        bare = ops.Handle(dev()).bind()
    check_names = not tunables and not workspace
    for k, v in tunables:
        ops.set_tunable(k, v)
    try:
        wm = fwd_pack(w, bool(tr))
        wf = torch.empty(wm.numel(), dtype=dt, device=dev())
        wd = torch.empty(wm.numel(), dtype=dt, device=dev())
        ops.pack_weights(dt, wm, Cout, 16, Cin, wf, wd)
        X1, X2, DY = nhwc(x1, dt), (nhwc(x2, dt) if C2 else None), nhwc(dy, dt)
        used = []
        # forward
        used.append(_named(ops, d, 0, names[0]) if check_names else ops.conv_kernel_name(d, 0))
        y = torch.empty(N * OH * OH * Cout, dtype=dt, device=dev())
        stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, device=dev())
        ops.conv_fwd(d, X1, X2, wf, bias.to(dev()), y_raw=y, stats=stats)
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(y, N, OH, OH, Cout), y_ref.bfloat16().float()), (name, "forward")
        rows = ops.conv_fwd_stats_rows(d)
        st = stats[:rows * 2 * Cout].view(rows, 2, Cout).double().sum(0).cpu()
        # BatchNorm partial sums come from the fp32 accumulators: integers, exact while the row sums stay below 2^24
        yd = y_ref.double()
        assert torch.equal(st[0], yd.sum((0, 2, 3))) or float((st[0] - yd.sum((0, 2, 3))).abs().max()) <= 1e-6 * float(yd.abs().sum()), name
        # input gradient, split over the two sources
        used.append(_named(ops, d, 1, names[1]) if check_names else ops.conv_kernel_name(d, 1))
        dx1 = torch.empty(N * H * H * C1, dtype=dt, device=dev())
        dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev()) if C2 else None
        ops.conv_dgrad(d, DY, wd, dx1, dx2)
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(dx1, N, H, H, C1), dx_ref[:, :C1].bfloat16().float()), (name, "dgrad x1")
        if C2:
            assert torch.equal(from_nhwc(dx2, N, H, H, C2), dx_ref[:, C1:].bfloat16().float()), (name, "dgrad x2")
        # weight gradient (fp32, accumulating)
        used.append(_named(ops, d, 2, names[2]) if check_names else ops.conv_kernel_name(d, 2))
        dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
        ops.conv_wgrad(d, X1, X2, DY, dw, None)
        torch.cuda.synchronize()
        assert torch.equal(unpack_fwd(dw, Cout, Cin, bool(tr)), dw_ref), (name, "wgrad")
        return used
    finally:
        for k, _ in tunables:
            ops.set_tunable(k)          # back to the built-in default
        if bare is not None:
            default.bind()
            bare.close()


@pytest.mark.parametrize("workspace", [False, True], ids=["named", "with_workspace"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_mfma_kernels_bit_exact_on_integer_data(pai, case, workspace):
    if workspace and case[0].startswith("cfg2_") and not isinstance(case[8][0], tuple):
        pytest.skip("full-size layers with thousands of tiles never split: covered by the named run")
    used = _run_case(pai, case, workspace=workspace)
    if workspace and case[0].endswith("_splitk"):
        # long-K / few-row layers run split over K once the workspace is registered (the cost model of fwd_cfg
        # decides for the mid-size ones: cfg2_enc4 forward splits, its input gradient does not)
        assert used[0] == SPLIT[0] and used[1] == SPLIT[0], used


# ---- the pipelined K loop of gg_wgrad_patch3_k: every way a pixel split can end ---------------------------------------
# The loop runs pairs of steps while two more follow and finishes with 1-3 steps whose look-ahead (next fragments, fill after
# next) is switched off one by one; a split of ONE step skips the second fill of the prologue.  (tr, N, H, C1, C2, Cout,
# relu, wgrad3_target): the split count the target gives makes the splits 1 .. 7 steps long, for the three forms of the
# kernel (128 x 64 wave tile, its 8 x 8-pixel form, 64 x 128)
PIPE_TAILS = [(0, 1, 32, 64, 0, 128, 0, 16), (0, 2, 32, 64, 0, 128, 0, 12), (0, 2, 32, 64, 0, 128, 0, 8), (0, 3, 32, 64, 0, 128, 0, 20),
              (0, 5, 32, 64, 0, 128, 0, 12), (0, 7, 32, 128, 0, 128, 0, 24),
              (1, 1, 16, 128, 128, 64, 1, 16), (1, 3, 16, 128, 128, 64, 1, 40), (1, 5, 16, 128, 128, 64, 1, 24),
              (0, 5, 16, 64, 0, 128, 0, 12), (1, 3, 8, 128, 0, 128, 1, 24)]


@pytest.mark.parametrize("cfg", PIPE_TAILS, ids=str)
def test_pipelined_weight_gradient_loop_on_every_split_length(pai, cfg):
    from thesis_pai_reconstruction_amd import ops
    tr, N, H, C1, C2, Cout, relu, target = cfg
    case = ("pipe_tail", tr, N, H, C1, C2, Cout, relu, None)
    for pipe in (1, 0):
        used = _run_case(pai, case, tunables=(("wgrad3_target", target), ("wgrad3_minrows", 64), ("wgrad3_pipe", pipe)))
        assert used[2].startswith("gg_wgrad_patch3_k"), used


OLD_WGRAD = {"enc_patch": "gg_wgrad_patch_k<128>", "dec_patch": "gg_wgrad_patch_k<128>", "dec_patch256x64": "gg_wgrad_patch_k<64>",
             "cfg2_dec5": "gg_wgrad_patch_k<128>"}


@pytest.mark.parametrize("name", sorted(OLD_WGRAD))
def test_previous_weight_gradient_kernels_stay_exact(pai, name):
    """gg_wgrad_patch_k (round 1-2; still the kernel of layers gg_wgrad_patch3_k does not take, e.g. 32-channel
    multiples): tunable wgrad3 = 0 routes the same cases back to it."""
    from thesis_pai_reconstruction_amd import ops
    case = next(c for c in CASES if c[0] == name)
    used = _run_case(pai, case, tunables=(("wgrad3", 0),))
    assert used[2] == OLD_WGRAD[name], used
    ops.set_tunable("wgrad3", 0)
    ops.set_tunable("wgrad_slab", 0)
    try:
        d = ops.make_desc(torch.bfloat16, case[1], case[2], case[3], case[3], case[4], case[5], case[6], 2, 0, 0, ops.ACT_NONE)
        assert ops.conv_wgrad_workspace_bytes(d) >= 0
    finally:
        ops.set_tunable("wgrad3")
        ops.set_tunable("wgrad_slab")


@pytest.mark.parametrize("slab", [1, 0], ids=["slabs", "atomics"])
@pytest.mark.parametrize("name", ["enc_patch256", "dec_patch256x64", "cfg2_D1", "cfg2_dec4", "cfg2_enc4", "cfg2_dec3"])
def test_weight_gradient_bias_and_overwrite(pai, name, slab):
    """gg_wgrad_patch3_k with a bias gradient (column sums of dY taken from the fragments every wave holds, spread over
    the workgroups that share a dY tile) and through pai_conv_wgrad_overwrite (dW / dbias need no zero fill: the slab sum,
    or the single writer of an un-split tile, stores them) -- exact on integer data, with the pixel splits meeting through
    slabs and through fp32 atomics."""
    from thesis_pai_reconstruction_amd import ops
    case = next(c for c in CASES if c[0] == name)
    _, tr, N, H, C1, C2, Cout, relu, names = case
    Cin, dt = C1 + C2, torch.bfloat16
    x1 = _ints((N, C1, H, H), 1)
    x2 = _ints((N, C2, H, H), 2) if C2 else None
    OH = H * 2 if tr else H // 2
    dy = _ints((N, Cout, OH, OH), 5)
    w = torch.zeros((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4))
    _, _, dw_ref = _reference(tr, x1, x2, w, dy, relu)
    db_ref = dy.sum((0, 2, 3))
    assert float(dw_ref.abs().max()) < 2 ** 24 and float(db_ref.abs().max()) < 2 ** 24
    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, relu, relu if C2 else 0, ops.ACT_NONE)
    ops.ensure_wgrad_workspace([d], dev())
    ops.set_tunable("wgrad_slab", slab)
    try:
        assert ops.conv_kernel_name(d, 2) == names[2]
        X1, X2, DY = nhwc(x1, dt), (nhwc(x2, dt) if C2 else None), nhwc(dy, dt)
        n = Cout * 16 * Cin
        for overwrite in (0, 1, 2):
            # 0 accumulate: starts from a known integer offset; 1 overwrite: starts from garbage; 2 pai_conv_wgrad_overwrite_w:
            # the weights start from garbage, the bias gradient is added
            dw = torch.full((n,), 3.0, device=dev()) if not overwrite else torch.full((n,), float("nan"), device=dev())
            db = torch.full((Cout,), 5.0, device=dev()) if overwrite != 1 else torch.full((Cout,), float("nan"), device=dev())
            (ops.conv_wgrad, ops.conv_wgrad_overwrite, ops.conv_wgrad_overwrite_w)[overwrite](d, X1, X2, DY, dw, db)
            torch.cuda.synchronize()
            off_w, off_b = (0.0 if overwrite else 3.0), (0.0 if overwrite == 1 else 5.0)
            assert torch.equal(unpack_fwd(dw, Cout, Cin, bool(tr)), dw_ref + off_w), (name, "dw", overwrite)
            assert torch.equal(db.cpu(), db_ref + off_b), (name, "dbias", overwrite)
    finally:
        ops.set_tunable("wgrad_slab")


def test_two_handles_keep_their_own_buffers(pai):
    """workspace and scratch belong to a per-device handle.  Two handles of one device, bound in turn,
    run the same split-K layer into their OWN workspaces with identical results; a handle without a workspace runs the
    layer un-split (other kernel, same bits on integer data)."""
    from thesis_pai_reconstruction_amd import ops
    case = next(c for c in CASES if c[0] == "enc_splitk")
    name, tr, N, H, C1, C2, Cout, relu, _ = case
    dt = torch.bfloat16
    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, 0, 0, ops.ACT_NONE)
    x1, w = _ints((N, C1, H, H), 1), _ints((Cout, C1, 4, 4), 3)
    y_ref = F.conv2d(x1, w, None, stride=2, padding=1).bfloat16().float()
    wm = fwd_pack(w, False)
    wf = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, C1, wf, None)
    X1 = nhwc(x1, dt)
    default = ops.handle_for(dev())
    ha, hb, hc = ops.Handle(dev()), ops.Handle(dev()), ops.Handle(dev())
    try:
        need = ops.conv_workspace_bytes(d, 0)
        ha.ensure_workspace(need)
        hb.ensure_workspace(need)
        assert ha.workspace.data_ptr() != hb.workspace.data_ptr()
        outs = []
        for h, split in ((ha, True), (hb, True), (hc, False)):
            h.bind()
            assert ops.conv_kernel_name(d, 0) == (SPLIT[0] if split else SPLIT[1])
            if split:
                h.workspace.fill_(float("nan"))          # the split launches overwrite their slabs, never accumulate
            y = torch.empty(N * (H // 2) ** 2 * Cout, dtype=dt, device=dev())
            ops.conv_fwd(d, X1, None, wf, None, y_raw=y)
            torch.cuda.synchronize()
            outs.append(from_nhwc(y, N, H // 2, H // 2, Cout))
        assert all(torch.equal(o, y_ref) for o in outs)
        # each split run wrote only its own handle's workspace
        assert not torch.isnan(ha.workspace[: need // 4]).all() and not torch.isnan(hb.workspace[: need // 4]).all()
    finally:
        default.bind()
        for h in (ha, hb, hc):
            h.close()


# ---- the fused producer backward of the input-gradient store (pai_conv_dgrad_bn) -----------------------------------
# (case of CASES, act1, act2, affine): decoder form (producer read through ReLU, no second gradient) and encoder form
# (LeakyReLU on the encoder path + the skip decoder's gradient through ReLU), on the kernels the benchmark runs
DGRAD_BN = [("dec_patch256", 2, 0, True), ("enc_dgrad256", 1, 2, True), ("cfg2_enc2", 1, 2, True), ("dec_splitk", 2, 0, True),
            ("enc_patch256", 1, 0, False)]


@pytest.mark.parametrize("cfg", DGRAD_BN, ids=[c[0] for c in DGRAD_BN])
def test_fused_producer_backward_bit_exact(pai, cfg):
    """pai_conv_dgrad_bn on integer data: du = act1'(pre) * g + act2'(pre) * add with pre = z * scale + shift, g the
    bf16-rounded input gradient; the stored du and (for ReLU producers, whose du stay integers) the BatchNorm-backward
    partial sums must equal the PyTorch-CPU restatement BIT FOR BIT -- the round-2 tests only held this store to its own
    two-pass form.  reference models/pix2pix.py:63-70,99-106 (the BatchNorm / activation backward aten runs as its own
    kernels)."""
    from thesis_pai_reconstruction_amd import ops
    name, act1, act2, affine = cfg
    case = next(c for c in CASES if c[0] == name)
    _, tr, N, H, C1, C2, Cout, relu, _ = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3)
    dy = _ints((N, Cout, OH, OH), 5)
    # input gradient w.r.t. the layer's (concatenated) input, exact integers
    x = torch.zeros(N, Cin, H, H, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=2, padding=1) if tr else F.conv2d(x, w, None, stride=2, padding=1)
    y.backward(dy)
    g = x.grad[:, :C1].bfloat16().float()                      # what the kernel holds when the store runs
    z = _ints((N, C1, H, H), 11, -3, 3)
    add = _ints((N, C1, H, H), 12) if act2 else None
    scale = torch.tensor([1.0, 2.0, 0.5, -1.0]).repeat(C1 // 4) if affine else None
    shift = torch.tensor([0.0, 1.0, -1.0, 2.0, -2.0, 0.5, 3.0, -0.5]).repeat(C1 // 8) if affine else None
    mean = torch.tensor([1.0, -1.0, 0.0, 2.0]).repeat(C1 // 4)
    rstd = torch.tensor([0.5, 1.0, 2.0, 0.25]).repeat(C1 // 4)
    pre = z * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) if affine else z
    pos = pre > 0

    def sel(v, act):
        return torch.where(pos, v, 0.2 * v if act == 1 else torch.zeros_like(v)) if act else v
    du = sel(g, act1) + (sel(add, act2) if add is not None else 0.0)
    du_bf = du.bfloat16().float()

    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, 0, 0, ops.ACT_NONE)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    wm = fwd_pack(w, bool(tr))
    wd = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, Cin, None, wd)
    dx1 = torch.empty(N * H * H * C1, dtype=dt, device=dev())
    dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev()) if C2 else None
    part = torch.full((ops.conv_dgrad_bn_rows_max(d) * 2 * C1,), float("nan"), device=dev())
    f = lambda t: None if t is None else t.to(dev())
    rows = ops.conv_dgrad_bn(d, nhwc(dy, dt), wd, dx1, dx2, nhwc(z, dt), act1, nhwc(add, dt) if add is not None else None, act2,
                             f(scale), f(shift), f(mean), f(rstd), part)
    torch.cuda.synchronize()
    assert ops.conv_kernel_id(d, 1) in (2, 3) and rows > 0                  # the matrix-core kernels: the store IS fused
    assert torch.equal(from_nhwc(dx1, N, H, H, C1), du_bf), (name, "du")
    if C2:
        assert torch.equal(from_nhwc(dx2, N, H, H, C2), x.grad[:, C1:].bfloat16().float()), (name, "skip half")
    P = part[: rows * 2 * C1].view(rows, 2, C1).double().sum(0).cpu()
    s1 = du_bf.double().sum((0, 2, 3))
    s2 = (du_bf.double() * ((z.double() - mean.view(1, -1, 1, 1).double()) * rstd.view(1, -1, 1, 1).double())).sum((0, 2, 3))
    if act1 != 1:       # integers all the way: exact
        assert torch.equal(P[0], s1) and torch.equal(P[1], s2), name
    else:               # 0.2 * g rounded to bf16 is no integer: fp32 sums in the tile's order
        assert float((P[0] - s1).abs().max()) <= 1e-5 * float(du_bf.abs().double().sum((0, 2, 3)).max())
        assert float((P[1] - s2).abs().max()) <= 1e-5 * float((du_bf.abs() * 8).double().sum((0, 2, 3)).max())


@pytest.mark.parametrize("name", ["enc_patch256", "dec_patch256", "enc_splitk"])
def test_relu_epilogue_of_minus_infinity_is_zero(pai, name):
    """The forward epilogue's activation (y_act = act(conv + bias)) on a -inf pre-activation: aten's relu gives 0 and
    LeakyReLU -inf; the branch-free form v > 0 ? v : v x slope would make the ReLU a NaN (checked on the chip with
    scripts/micro/inf_probe.hip).  One +inf input pixel under negative weights puts -inf into every output it reaches."""
    from thesis_pai_reconstruction_amd import ops
    case = next(c for c in CASES if c[0] == name)
    _, tr, N, H, C1, C2, Cout, relu, _ = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3, -2, -1)
    x = _ints((N, Cin, H, H), 1, 0, 1)
    x[0, 0, H // 2, H // 2] = float("inf")
    x[1, 0, 1, 2] = float("nan")                       # and a NaN stays a NaN through both activations (aten)
    bias = _ints((Cout,), 7, -3, 3)
    pre = F.conv_transpose2d(x, w, bias, stride=2, padding=1) if tr else F.conv2d(x, w, bias, stride=2, padding=1)
    assert (pre == float("-inf")).any() and not torch.isnan(pre[0]).any() and torch.isnan(pre[1]).any()
    wm = fwd_pack(w, bool(tr))
    wf = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, Cin, wf, None)
    X1 = nhwc(x[:, :C1], dt)
    X2 = nhwc(x[:, C1:], dt) if C2 else None
    for act, fn in ((ops.ACT_RELU, F.relu), (ops.ACT_LRELU, lambda v: F.leaky_relu(v, 0.2))):
        d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, 0, 0, act)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
        ya = torch.empty(N * OH * OH * Cout, dtype=dt, device=dev())
        ops.conv_fwd(d, X1, X2, wf, bias.to(dev()), y_act=ya)
        torch.cuda.synchronize()
        got, want = from_nhwc(ya, N, OH, OH, Cout), fn(pre).bfloat16().float()
        assert torch.equal(torch.isnan(got), torch.isnan(want)), (name, act, int(torch.isnan(got).sum()), int(torch.isnan(want).sum()))
        assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(want, nan=7.0)), (name, act)


@pytest.mark.parametrize("name", ["dec_patch256", "dec_splitk"])
def test_infinite_gradient_behind_a_relu_producer_is_zero(pai, name):
    """threshold_backward (the ReLU backward aten runs for reference models/pix2pix.py:99-106) SELECTS: an infinite gradient
    at a position whose pre-activation is <= 0 gives 0, not inf x 0 = NaN.  The fused store computes the negative side as
    g x slope for LeakyReLU; for ReLU it must select a literal zero.  One +inf in dy, positive weights (no inf - inf)."""
    from thesis_pai_reconstruction_amd import ops
    case = next(c for c in CASES if c[0] == name)
    _, tr, N, H, C1, C2, Cout, relu, _ = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3, 1, 2)
    dy = torch.zeros(N, Cout, OH, OH)
    dy[0, 1, OH // 2, OH // 2 + 1] = float("inf")
    dy[N - 1, 0, 0, 0] = 3.0
    x = torch.zeros(N, Cin, H, H, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=2, padding=1) if tr else F.conv2d(x, w, None, stride=2, padding=1)
    y.backward(dy)
    g = x.grad[:, :C1]
    assert torch.isinf(g).any() and not torch.isnan(g).any()
    z = _ints((N, C1, H, H), 11, -3, 3)
    want = torch.where(z > 0, g, torch.zeros_like(g)).bfloat16().float()
    assert torch.isinf(want).any() and (torch.isinf(g) & (z <= 0)).any()

    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, 0, 0, ops.ACT_NONE)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    wd = torch.empty(w.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, fwd_pack(w, bool(tr)), Cout, 16, Cin, None, wd)
    dx1 = torch.empty(N * H * H * C1, dtype=dt, device=dev())
    dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev()) if C2 else None
    part = torch.empty(ops.conv_dgrad_bn_rows_max(d) * 2 * C1, device=dev())
    mean, rstd = torch.zeros(C1, device=dev()), torch.ones(C1, device=dev())
    rows = ops.conv_dgrad_bn(d, nhwc(dy, dt), wd, dx1, dx2, nhwc(z, dt), ops.ACT_RELU, None, ops.ACT_NONE, None, None, mean, rstd, part)
    torch.cuda.synchronize()
    assert rows > 0
    got = from_nhwc(dx1, N, H, H, C1)
    assert not torch.isnan(got).any(), int(torch.isnan(got).sum())
    assert torch.equal(got, want), name


@pytest.mark.parametrize("shape", [(64, 128), (3, 16)], ids=["cfg2_dec7", "small"])
def test_fused_producer_backward_of_the_thin_head_bit_exact(pai, shape):
    """pai_conv_dgrad_bn on the head (ConvTranspose2d(64|64 -> 1), reference models/pix2pix.py:185-193): the input
    gradient runs on thin_fwd2_k and the first BatchNorm-backward pass of decoders[6] (read without an activation: du is
    the gradient itself) rides on its store.  Integer data: both halves of the gradient and the partial sums
    (sum du, sum du * xhat per channel, one row per workgroup) equal the PyTorch-CPU restatement bit for bit; with the
    fusion switched off (tunable thin_bwd = 0) the entry point's two-pass form gives the same du and the same sums."""
    from thesis_pai_reconstruction_amd import ops
    N, H = shape
    C1 = C2 = 64
    dt = torch.bfloat16
    w = _ints((C1 + C2, 1, 4, 4), 3)
    dy = _ints((N, 1, 2 * H, 2 * H), 5)
    x = torch.zeros(N, C1 + C2, H, H, requires_grad=True)
    F.conv_transpose2d(x, w, None, stride=2, padding=1).backward(dy)
    g = x.grad.bfloat16().float()
    z = _ints((N, C1, H, H), 11, -3, 3)
    mean = torch.tensor([1.0, -1.0, 0.0, 2.0]).repeat(C1 // 4)
    rstd = torch.tensor([0.5, 1.0, 2.0, 0.25]).repeat(C1 // 4)
    s1 = g[:, :C1].double().sum((0, 2, 3))
    s2 = (g[:, :C1].double() * ((z.double() - mean.view(1, -1, 1, 1).double()) * rstd.view(1, -1, 1, 1).double())).sum((0, 2, 3))
    d = ops.make_desc(dt, 1, N, H, H, C1, C2, 1, 2, 0, 0, ops.ACT_NONE)
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    wd = torch.empty(w.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, fwd_pack(w, True), 1, 16, C1 + C2, None, wd)
    f = lambda t: t.to(dev())
    for fused in (1, 0):
        ops.set_tunable("thin_bwd", fused)
        try:
            dx1 = torch.empty(N * H * H * C1, dtype=dt, device=dev())
            dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev())
            part = torch.full((ops.conv_dgrad_bn_rows_max(d) * 2 * C1,), float("nan"), device=dev())
            rows = ops.conv_dgrad_bn(d, nhwc(dy, dt), wd, dx1, dx2, nhwc(z, dt), ops.ACT_NONE, None, ops.ACT_NONE, None, None,
                                     f(mean), f(rstd), part)
            torch.cuda.synchronize()
        finally:
            ops.set_tunable("thin_bwd")
        assert ops.conv_kernel_id(d, 1) == 4 and rows > 0, (fused, rows)
        if fused:
            assert rows == min((N * H * H + 63) // 64, 4096), rows          # one partial row per workgroup of thin_fwd2_k
        assert torch.equal(from_nhwc(dx1, N, H, H, C1), g[:, :C1]) and torch.equal(from_nhwc(dx2, N, H, H, C2), g[:, C1:]), fused
        P = part[: rows * 2 * C1].view(rows, 2, C1).double().sum(0).cpu()
        assert torch.equal(P[0], s1) and torch.equal(P[1], s2), fused


# ---- the thin layers at the benchmark's own shapes (BASELINE configs[1]) --------------------------------------------
# (name, transposed, N, H, C1, C2, Cout): encoders[0], discriminator block 0 at the 2 x 64 batch of its own phase, and
# decoders[7] (reference models/pix2pix.py:141-147,185-193, models/wrapper.py:229)
THIN = [("cfg2_enc0", 0, 64, 256, 1, 0, 64), ("cfg2_D0", 0, 128, 256, 1, 1, 64), ("cfg2_dec7", 1, 64, 128, 64, 64, 1)]


@pytest.mark.parametrize("case", THIN, ids=[c[0] for c in THIN])
def test_thin_layers_bit_exact_at_config2_shapes(pai, case):
    """thin_fwd_k, thin_dgrad_gemm_k + thin_col2im_k and thin_wgrad_k on integer data at full size: every partial sum is
    an exact fp32 integer, so forward (+ bias, + LeakyReLU copy), input gradient and weight / bias gradient must equal
    PyTorch-CPU bit for bit -- byte-permute gathers, edge-lane masks and the col2im tap tables included.  The family is
    asserted strictly (4 = thin matrix-core kernels) with the scratch registered."""
    from thesis_pai_reconstruction_amd import ops
    name, tr, N, H, C1, C2, Cout = case
    Cin, dt = C1 + C2, torch.bfloat16
    OH = H * 2 if tr else H // 2
    x1 = _ints((N, C1, H, H), 1)
    x2 = _ints((N, C2, H, H), 2) if C2 else None
    w = _ints((Cin, Cout, 4, 4) if tr else (Cout, Cin, 4, 4), 3)
    bias = _ints((Cout,), 7, -3, 3)
    dy = _ints((N, Cout, OH, OH), 5)
    y_ref, dx_ref, dw_ref = _reference(tr, x1, x2, w, dy, 0, bias)
    db_ref = dy.sum((0, 2, 3))
    assert max(float(t.abs().max()) for t in (y_ref, dx_ref, dw_ref, db_ref)) < 2 ** 24
    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, 0, 0, ops.ACT_LRELU if not tr else ops.ACT_NONE)
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    want_ops = (0, 1, 2) if tr else (0, 2)
    assert all(ops.conv_kernel_id(d, op) == 4 for op in want_ops), [ops.conv_kernel_id(d, op) for op in (0, 1, 2)]
    wm = fwd_pack(w, bool(tr))
    wf = torch.empty(wm.numel(), dtype=dt, device=dev())
    wd = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, Cout, 16, Cin, wf, wd)
    X1, X2, DY, B = nhwc(x1, dt), (nhwc(x2, dt) if C2 else None), nhwc(dy, dt), bias.to(dev())
    if tr:      # decoders[7]: fp32 output (the engine adds tanh; the exact test leaves it out), both source gradients
        y32 = torch.empty(N * OH * OH * Cout, dtype=torch.float32, device=dev())
        ops.conv_fwd(d, X1, X2, wf, B, y_f32=y32)
        torch.cuda.synchronize()
        assert torch.equal(y32.cpu().view(N, OH, OH, Cout).permute(0, 3, 1, 2), y_ref), (name, "forward")
        dx1 = torch.empty(N * H * H * C1, dtype=dt, device=dev())
        dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev())
        ops.conv_dgrad(d, DY, wd, dx1, dx2)
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(dx1, N, H, H, C1), dx_ref[:, :C1].bfloat16().float()), (name, "dgrad x1")
        assert torch.equal(from_nhwc(dx2, N, H, H, C2), dx_ref[:, C1:].bfloat16().float()), (name, "dgrad x2")
    else:
        y = torch.empty(N * OH * OH * Cout, dtype=dt, device=dev())
        ya = torch.empty_like(y)
        ops.conv_fwd(d, X1, X2, wf, B, y_raw=y, y_act=ya)
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(y, N, OH, OH, Cout), y_ref.bfloat16().float()), (name, "forward")
        assert torch.equal(from_nhwc(ya, N, OH, OH, Cout), F.leaky_relu(y_ref, 0.2).bfloat16().float()), (name, "forward act")
        del y, ya
        if C2:  # discriminator block 0: the gradient w.r.t. the image half only (what the generator phase asks for)
            dx2 = torch.empty(N * H * H * C2, dtype=dt, device=dev())
            ops.conv_dgrad(d, DY, wd, None, dx2, only_c2=True)
            torch.cuda.synchronize()
            assert torch.equal(from_nhwc(dx2, N, H, H, C2), dx_ref[:, C1:].bfloat16().float()), (name, "dgrad x2")
    dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
    db = torch.zeros(Cout, dtype=torch.float32, device=dev())
    ops.conv_wgrad(d, X1, X2, DY, dw, db)
    torch.cuda.synchronize()
    assert torch.equal(unpack_fwd(dw, Cout, Cin, bool(tr)), dw_ref), (name, "wgrad")
    assert torch.equal(db.cpu(), db_ref), (name, "dbias")


# stride-1 "same" convolutions of the residual / Trans U-Nets at their configs[3] / configs[4] layer shapes (batch reduced
# where the tensors would not fit a test): (k, N, H, W, C1, C2, Cout, kernels that may serve forward / input gradient / weight
# gradient -- the family is asserted, the split choice is the library's)
TILE = ("gg_fwd_mfma_k<128, 128, false, false, 64>", "gg_fwd_mfma_k<128, 64, false, false, 64>",
        "gg_fwd_mfma_k<128, 128, true, true, 64>", "gg_fwd_mfma_k<128, 64, true, false, 64>")
WG = ("gg_wgrad_mfma_k<128>", "gg_wgrad_mfma_k<64>")
SAME_CASES = [
    # ResidualBlockNeXt 1 x 1 up / down projections and skips at the fine levels: the streaming pointwise kernel (gg_pw.hip)
    ("res_1x1_64_128", 1, 4, 256, 256, 64, 0, 128, ("pwx_k<64, 128>",), ("pwx_k<128, 64>",), WG),
    ("res_1x1_128_64", 1, 4, 256, 256, 128, 0, 64, ("pwx_k<128, 64>",), ("pwx_k<64, 128>",), WG),
    ("res_1x1_cat", 1, 4, 128, 128, 64, 64, 128, ("pwx_k<128, 128>",), ("pwx_k<128, 128>",), WG),   # decoder: two sources read as one concatenation
    ("res_1x1_cat_64", 1, 4, 128, 128, 64, 64, 64, ("pwx_k<128, 64>",), ("pwx_k<64, 128>",), WG),   # its skip convolution
    ("res_1x1_256_64", 1, 2, 128, 128, 128, 128, 64, ("pwx_k<256, 64>",), ("pwx_k<64, 256>",), WG), # level-1 decoder skip (128 | 128 -> 64)
    ("res_1x1_64_64", 1, 2, 128, 128, 64, 0, 64, ("pwx_k<64, 64>",), ("pwx_k<64, 64>",), WG),
    # 128 x 256 / 256 x 128 filters: two waves of a workgroup take half of the output channels each
    ("res_1x1_128_256", 1, 2, 128, 128, 128, 0, 256, ("pwx_k<128, 256>",), ("pwx_k<256, 128>",), WG),
    ("res_1x1_cat_256_128", 1, 2, 128, 128, 128, 128, 128, ("pwx_k<256, 128>",), ("pwx_k<128, 256>",), WG),
    ("res_1x1_ragged", 1, 5, 100, 37, 64, 64, 128, ("pwx_k<128, 128>",), ("pwx_k<128, 128>",), WG + ("gg_simt",)),   # 18500 pixels: partial group, partial batch
    ("res_1x1_small", 1, 4, 32, 32, 64, 0, 128, TILE, TILE, WG),              # < 16384 pixels: the tile kernels
    ("tr_3x3_cat", 3, 8, 128, 128, 64, 64, 64, TILE, TILE, WG),               # TransUNet decoder block, configs[4] width
    ("tr_3x3_128", 3, 8, 32, 32, 128, 128, 128, TILE, TILE, WG),
    ("tr_3x3_16", 3, 4, 256, 256, 16, 0, 16, ("small_mfma_bf16",), ("small_mfma_bf16",), WG),      # 16-channel encoder block
    ("gate_1x1_64_32", 1, 8, 128, 128, 64, 0, 32, ("thin_mfma_bf16",), ("thin_mfma_bf16",), WG),   # AttentionBlock W_x / W_g (pw_k)
    ("gate_1x1_128_64", 1, 8, 64, 64, 128, 0, 64, ("pwx_k<128, 64>",), ("pwx_k<64, 128>",), WG),
    ("in_conv_3x3", 3, 4, 256, 256, 1, 0, 64, ("thin_mfma_bf16",), None, ("thin_mfma_bf16",)),   # 1 -> 64 in_conv (no dgrad)
    ("out_conv_3x3", 3, 4, 256, 256, 64, 0, 1, ("thin_mfma_bf16",), ("thin_mfma_bf16",), ("thin_mfma_bf16",)),
]


@pytest.mark.parametrize("case", SAME_CASES, ids=[c[0] for c in SAME_CASES])
def test_same_convolutions_of_the_other_families_bit_exact(pai, case):
    """nn.Conv2d(k = 1 | 3, padding = k // 2) of reference models/res_unet.py:59-62,86-95,265,308 and models/trans_unet.py:66,98,
    203-227 on the kernels the composable networks run (tile, small-channel and thin families): forward with bias, both
    input gradients and the weight / bias gradient against PyTorch-CPU on small-integer data, bit for bit."""
    from thesis_pai_reconstruction_amd import ops
    name, k, N, H, W, C1, C2, K, kf, kd, kw = case
    dtype = torch.bfloat16
    Cin = C1 + C2
    x1 = _ints((N, C1, H, W), 1).requires_grad_(True)
    x2 = _ints((N, C2, H, W), 2).requires_grad_(True) if C2 else None
    w = _ints((K, Cin, k, k), 3).requires_grad_(True)
    b = _ints((K,), 4, -8, 8).requires_grad_(True)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    y = F.conv2d(xin, w, b, padding=k // 2)
    dy = _ints(tuple(y.shape), 5)
    y.backward(dy)
    d = ops.make_desc(dtype, 0, N, H, W, C1, C2, K, 1, 0, 0, ops.ACT_NONE, kernel=k)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    ops.ensure_wgrad_workspace([d], dev())
    for op, want in ((0, kf), (1, kd), (2, kw)):
        if want is not None:
            _named(ops, d, op, want)
    wm = w.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    wd = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, K, k * k, Cin, wf, wd)
    X1, DY = nhwc(x1.detach(), dtype), nhwc(dy, dtype)
    X2 = nhwc(x2.detach(), dtype) if C2 else None
    if K > 2:
        yo = torch.full((N * H * W * K,), 7.0, dtype=dtype, device=dev())
        ops.conv_fwd(d, X1, X2, wf, b.detach().to(dev()), y_raw=yo)
        got_y = from_nhwc(yo, N, H, W, K)
        # the same launch with BatchNorm partial statistics (fp32 accumulators: integers)
        yo2 = torch.full_like(yo, 7.0)
        stats = torch.full((ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * K,), float("nan"), device=dev())
        ops.conv_fwd(d, X1, X2, wf, b.detach().to(dev()), y_raw=yo2, stats=stats)
        torch.cuda.synchronize()
        assert torch.equal(yo2, yo), name
        rows = ops.conv_fwd_stats_rows(d)
        st = stats[:rows * 2 * K].view(rows, 2, K).double().sum(0).cpu()
        yd = y.detach().double()
        assert float((st[0] - yd.sum((0, 2, 3))).abs().max()) <= 1e-6 * float(yd.abs().sum()), name
        assert float((st[1] - (yd * yd).sum((0, 2, 3))).abs().max()) <= 1e-6 * float((yd * yd).sum()), name
    else:       # 1-channel outputs leave in fp32 (the head of the network)
        yo = torch.full((N * H * W * K,), 7.0, dtype=torch.float32, device=dev())
        ops.conv_fwd(d, X1, X2, wf, b.detach().to(dev()), y_f32=yo)
        got_y = from_nhwc(yo, N, H, W, K)
    torch.cuda.synchronize()
    want_y = y.detach().to(dtype).float() if K > 2 else y.detach()
    assert torch.equal(got_y, want_y), name
    if kd is not None:
        dx1 = torch.full((N * H * W * C1,), 7.0, dtype=dtype, device=dev())
        dx2 = torch.full((N * H * W * C2,), 7.0, dtype=dtype, device=dev()) if C2 else None
        ops.conv_dgrad(d, DY, wd, dx1, dx2)
        torch.cuda.synchronize()
        assert torch.equal(from_nhwc(dx1, N, H, W, C1), x1.grad.to(dtype).float()), name
        if C2:
            assert torch.equal(from_nhwc(dx2, N, H, W, C2), x2.grad.to(dtype).float()), name
    dw = torch.full((wm.numel(),), float("nan"), dtype=torch.float32, device=dev())
    db = torch.full((K,), float("nan"), dtype=torch.float32, device=dev())
    ops.conv_wgrad_overwrite(d, X1, X2, DY, dw, db)
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu().view(K, k, k, Cin).permute(0, 3, 1, 2), w.grad), name
    assert torch.equal(db.cpu(), b.grad), name



# (N, H, W, Cin, Cout, k, groups): layers of the ResNeXt blocks that read their input through a prologue -- the pointwise
# ones (pwx_k / gg_wgrad_mfma_k) and the grouped 3 x 3 (grouped3_k / grouped3_wgrad_k: zero padding must stay zero)
PRO_CASES = [(4, 128, 128, 128, 64, 1, 1), (2, 128, 128, 128, 128, 1, 1), (4, 64, 64, 64, 128, 1, 1), (2, 128, 64, 64, 256, 1, 1),
             (1, 128, 128, 256, 64, 1, 1), (2, 64, 64, 128, 128, 3, 32), (1, 16, 32, 128, 128, 3, 32), (1, 128, 128, 128, 256, 1, 1),
             (1, 128, 128, 256, 128, 1, 1)]


@pytest.mark.parametrize("act", ["relu", "none"])
@pytest.mark.parametrize("case", PRO_CASES, ids=str)
def test_input_prologue_equals_batchnorm_pass_then_plain_call(pai, case, act):
    """pai_conv_fwd_pro / pai_conv_wgrad_pro (Conv2d -> BatchNorm2d -> ReLU -> Conv2d of reference models/res_unet.py:143-147
    without the activated tensor in the middle) against pai_bn_apply followed by the plain calls, on random data: the prologue
    forms the same bf16 values on load, so outputs and BatchNorm partial statistics are BIT-IDENTICAL and the weight / bias gradients equal up to the order of their fp32 atomics."""
    from thesis_pai_reconstruction_amd import ops
    N, H, W, Cin, K, k, groups = case
    dt, M = torch.bfloat16, N * H * W
    d = ops.make_desc(dt, 0, N, H, W, Cin, 0, K, 1, 0, 0, ops.ACT_NONE, kernel=k, groups=groups)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    ops.ensure_wgrad_workspace([d], dev())
    assert ops.conv_prologue_ok(d)
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(dt).to(dev())
    dy = torch.randn(M, K, generator=g).to(dt).to(dev())
    w = (torch.randn(K * k * k * Cin, generator=g) * 0.1).to(dt).to(dev())
    bias = torch.randn(K, generator=g).to(dev())
    scale = (0.5 + torch.rand(Cin, generator=g)).to(dev())
    shift = (torch.randn(Cin, generator=g) * 0.3).to(dev())
    a_code = ops.ACT_RELU if act == "relu" else ops.ACT_NONE
    rows = ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * K
    # reference: the BatchNorm + activation as a pass of its own
    a = torch.empty_like(z)
    ops.bn_apply(dt, z, M, Cin, scale, shift, a_code, a)
    y0, st0 = torch.empty(M, K, dtype=dt, device=dev()), torch.zeros(rows, device=dev())
    ops.conv_fwd(d, a, None, w, bias, y_raw=y0, stats=st0)
    nw = K * k * k * Cin
    with_db = groups == 1                 # (the grouped layer feeds a BatchNorm: its kernel forms no bias gradient)
    dw0, db0 = torch.zeros(nw, device=dev()), (torch.empty(K, device=dev()) if with_db else None)
    ops.conv_wgrad_overwrite(d, a, None, dy, dw0, db0)
    # prologue
    y1, st1 = torch.empty(M, K, dtype=dt, device=dev()), torch.zeros(rows, device=dev())
    ops.conv_fwd_pro(d, z, w, bias, y1, st1, scale, shift, a_code)
    dw1, db1 = torch.zeros(nw, device=dev()), (torch.full((K,), float("nan"), device=dev()) if with_db else None)
    ops.conv_wgrad_pro(d, z, dy, dw1, db1, True, scale, shift, a_code)
    y2 = torch.empty(M, K, dtype=dt, device=dev())
    ops.conv_fwd_pro(d, z, w, bias, y2, None, scale, shift, a_code)          # without statistics
    dw2 = torch.ones(nw, device=dev())
    ops.conv_wgrad_pro(d, z, dy, dw2, None, False, scale, shift, a_code)     # accumulating
    torch.cuda.synchronize()
    n = ops.conv_fwd_stats_rows(d) * 2 * K
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
    assert torch.equal(y2.view(torch.int16), y0.view(torch.int16))
    assert torch.equal(st1[:n], st0[:n])
    # (the pixel splits of the weight gradient meet through fp32 atomics: equal up to their summation order)
    tol = 2e-6 * float(dw0.abs().max()) * 8
    assert float((dw1 - dw0).abs().max()) <= tol
    if with_db:
        assert float((db1 - db0).abs().max()) <= 2e-6 * float(db0.abs().max()) * 8
    else:
        assert torch.equal(dw1, dw0)          # the grouped kernel sums its partial blocks in a fixed order
    written = dw0 != 0 if groups > 1 else torch.ones_like(dw0, dtype=torch.bool)     # (diagonal blocks only)
    assert float(((dw2 - 1.0 - dw0) * written).abs().max()) <= tol
    # and against fp64 on the bf16 activation (the plain call's own tolerance)
    if k == 1:
        af = a.double()
        want = af @ w.double().view(K, Cin).t() + bias.double()
        assert float((y1.double() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())


def test_prologue_is_refused_where_no_kernel_takes_it(pai):
    from thesis_pai_reconstruction_amd import ops
    dt = torch.bfloat16
    for args, kw in (((dt, 0, 2, 32, 32, 128, 0, 64, 1, 0, 0, ops.ACT_NONE), dict(kernel=1)),        # < 16384 pixels
                     ((dt, 0, 2, 128, 128, 64, 64, 64, 1, 0, 0, ops.ACT_NONE), dict(kernel=1)),      # two sources
                     ((dt, 0, 2, 128, 128, 64, 0, 64, 1, 0, 0, ops.ACT_NONE), dict(kernel=3)),       # 3 x 3
                     ((torch.float32, 0, 2, 128, 128, 64, 0, 64, 1, 0, 0, ops.ACT_NONE), dict(kernel=1))):
        assert not ops.conv_prologue_ok(ops.make_desc(*args, **kw))


# (N, H, W, Cin, Cout): the input gradient of a pointwise layer (dx has Cin channels) with the producer's backward in the store
PWX_BWD = [(2, 128, 128, 128, 64), (2, 128, 128, 64, 128), (1, 128, 128, 128, 128), (5, 100, 37, 64, 128), (1, 128, 128, 64, 256),
           (1, 128, 128, 256, 128), (1, 128, 128, 128, 256)]


@pytest.mark.parametrize("act1", [2, 0], ids=["relu", "none"])
@pytest.mark.parametrize("case", PWX_BWD, ids=str)
def test_fused_producer_backward_of_the_streaming_pointwise_kernel_bit_exact(pai, case, act1):
    """pai_conv_dgrad_bn on pwx_k (the 1 x 1 behind the grouped 3 x 3 of a ResNeXt block hands its input gradient to that
    layer's BatchNorm, reference models/res_unet.py:143-147): du = act1'(z * scale + shift) * dgrad and the partial sums
    (sum du, sum du * xhat) on integer data against the PyTorch-CPU restatement, bit for bit."""
    from thesis_pai_reconstruction_amd import ops
    N, H, W, Cin, K = case
    dt = torch.bfloat16
    w = _ints((K, Cin, 1, 1), 3)
    dy = _ints((N, K, H, W), 5)
    x = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(x, w).backward(dy)
    g = x.grad.bfloat16().float()
    z = _ints((N, Cin, H, W), 11, -3, 3)
    scale = torch.tensor([1.0, 2.0, 0.5, -1.0]).repeat(Cin // 4)
    shift = torch.tensor([0.0, 1.0, -1.0, 2.0, -2.0, 0.5, 3.0, -0.5]).repeat(Cin // 8)
    mean = torch.tensor([1.0, -1.0, 0.0, 2.0]).repeat(Cin // 4)
    rstd = torch.tensor([0.5, 1.0, 2.0, 0.25]).repeat(Cin // 4)
    pre = z * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    du = torch.where(pre > 0, g, torch.zeros_like(g)) if act1 else g
    du_bf = du.bfloat16().float()
    d = ops.make_desc(dt, 0, N, H, W, Cin, 0, K, 1, 0, 0, ops.ACT_NONE, kernel=1)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
    assert ops.conv_kernel_name(d, 1) == f"pwx_k<{K}, {Cin}>"
    wm = w.permute(0, 2, 3, 1).contiguous().to(dev())
    wd = torch.empty(wm.numel(), dtype=dt, device=dev())
    ops.pack_weights(dt, wm, K, 1, Cin, None, wd)
    dx = torch.full((N * H * W * Cin,), 7.0, dtype=dt, device=dev())
    part = torch.full((ops.conv_dgrad_bn_rows_max(d) * 2 * Cin,), float("nan"), device=dev())
    f = lambda t: t.to(dev())
    rows = ops.conv_dgrad_bn(d, nhwc(dy, dt), wd, dx, None, nhwc(z, dt), act1, None, ops.ACT_NONE, f(scale), f(shift), f(mean),
                             f(rstd), part)
    torch.cuda.synchronize()
    assert 0 < rows <= 4096
    assert torch.equal(from_nhwc(dx, N, H, W, Cin), du_bf)
    P = part[: rows * 2 * Cin].view(rows, 2, Cin).double().sum(0).cpu()
    s1 = du_bf.double().sum((0, 2, 3))
    s2 = (du_bf.double() * ((z.double() - mean.view(1, -1, 1, 1).double()) * rstd.view(1, -1, 1, 1).double())).sum((0, 2, 3))
    assert torch.equal(P[0], s1) and torch.equal(P[1], s2)
    # without partial sums: the store alone
    dx2 = torch.full_like(dx, 7.0)
    assert ops.conv_dgrad_bn(d, nhwc(dy, dt), wd, dx2, None, nhwc(z, dt), act1, None, ops.ACT_NONE, f(scale), f(shift)) == 0
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
