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
    ("enc_patch", 0, 4, 128, 64, 0, 256, 0, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch_k<128>")),
    ("dec_patch", 1, 4, 64, 64, 64, 128, 1, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch_k<128>")),
    ("enc_patch256", 0, 8, 256, 64, 0, 128, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch_k<128>")),
    ("dec_patch256", 1, 8, 64, 128, 0, 128, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 128, true>", "gg_wgrad_patch_k<128>")),
    ("dec_patch256x64", 1, 8, 64, 128, 128, 64, 1, ("gg_fwd_patch_k<128, 64, false>", "gg_fwd_patch_k<128, 128, true>", "gg_wgrad_patch_k<64>")),
    ("enc_dgrad256", 0, 8, 128, 128, 0, 128, 0, ("gg_fwd_patch_k<128, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
    ("enc_splitk", 0, 4, 8, 256, 0, 256, 0, (SPLIT, SPLIT, "gg_wgrad_mfma_k<128>")),
    ("dec_splitk", 1, 4, 4, 256, 256, 256, 1, (SPLIT, SPLIT, "gg_wgrad_mfma_k<128>")),
    # BASELINE configs[1] layer shapes at the benchmark batch (64; the discriminator sees 2 x 64 in its own phase)
    ("cfg2_enc2", 0, 64, 64, 128, 0, 256, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
    ("cfg2_enc4", 0, 64, 16, 512, 0, 512, 0, (SPLIT, SPLIT, "gg_wgrad_mfma_k<128>")),
    ("cfg2_dec3", 1, 64, 8, 512, 512, 512, 1, (SPLIT, SPLIT, "gg_wgrad_mfma_k<128>")),
    ("cfg2_dec4", 1, 64, 16, 512, 512, 256, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
    ("cfg2_dec5", 1, 64, 32, 256, 256, 128, 1, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
    ("cfg2_dec6", 1, 64, 64, 128, 128, 64, 1, ("gg_fwd_patch1_k<256, 64, false>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<64>")),
    ("cfg2_D1", 0, 128, 128, 64, 0, 128, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<128, 64, false>", "gg_wgrad_patch_k<128>")),
    ("cfg2_D2", 0, 128, 64, 128, 0, 256, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
    ("cfg2_D3", 0, 128, 32, 256, 0, 512, 0, ("gg_fwd_patch_k<256, 128, true>", "gg_fwd_patch_k<256, 128, true>", "gg_wgrad_patch_k<128>")),
]


def _ints(shape, seed, lo=-2, hi=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _named(ops, d, op, want):
    got = ops.conv_kernel_name(d, op)
    assert got in ((want,) if isinstance(want, str) else want), (op, got, want)
    return got


def _reference(tr, x1, x2, w, dy, relu):
    """PyTorch-CPU fp32 (exact on this data): y, dx (w.r.t. the relu'd, concatenated input), dw."""
    xs = [F.relu(x1) if relu else x1]
    if x2 is not None:
        xs.append(F.relu(x2) if relu else x2)
    x = torch.cat(xs, 1).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv_transpose2d(x, wr, None, stride=2, padding=1) if tr else F.conv2d(x, wr, None, stride=2, padding=1)
    y.backward(dy)
    return y.detach(), x.grad, wr.grad


def _run_case(pai, case, tunables=(), workspace=True, frag=False):
    """frag=True: both filter packs carry their fragment-major copy (pai_pack_frag, pai_conv_desc.pack_flags = 3).
    workspace=False: a handle WITHOUT split-K workspace, the setting the kernel names of CASES were recorded in
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
    y_ref, dx_ref, dw_ref = _reference(tr, x1, x2, w, dy, relu)
    assert float(y_ref.abs().max()) < 2 ** 24 and float(dw_ref.abs().max()) < 2 ** 24     # exact in fp32

    d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, relu, relu if C2 else 0, ops.ACT_NONE)
    d.pack_flags = 3 if frag else 0
    default = ops.handle_for(dev())
    bare = None
    if workspace:
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev())
        ops.ensure_scratch(ops.scratch_bytes_for([d]), dev())
    else:
        bare = ops.Handle(dev()).bind()
    check_names = not tunables and not workspace
    lib = pai.lib.load()
    for k, v in tunables:
        assert lib.pai_set_tunable(k.encode(), v) == 0
    try:
        wm = fwd_pack(w, bool(tr))
        wf = torch.empty(wm.numel() * (2 if frag else 1), dtype=dt, device=dev())
        wd = torch.empty(wm.numel() * (2 if frag else 1), dtype=dt, device=dev())
        ops.pack_weights(dt, wm, Cout, 16, Cin, wf, wd)
        if frag:
            ops.pack_frag(wf, Cout, 16 * Cin, wf[wm.numel():])
            ops.pack_frag(wd, Cin, 16 * Cout, wd[wm.numel():])
        X1, X2, DY = nhwc(x1, dt), (nhwc(x2, dt) if C2 else None), nhwc(dy, dt)
        used = []
        # forward
        used.append(_named(ops, d, 0, names[0]) if check_names else ops.conv_kernel_name(d, 0))
        y = torch.empty(N * OH * OH * Cout, dtype=dt, device=dev())
        stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, device=dev())
        ops.conv_fwd(d, X1, X2, wf, None, y_raw=y, stats=stats)
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
            lib.pai_set_tunable(k.encode(), 0)
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


P2_CASES = [c for c in CASES if c[0] in ("enc_patch256", "dec_patch256", "enc_dgrad256", "cfg2_enc2", "cfg2_dec4", "cfg2_dec5", "cfg2_D3")]


@pytest.mark.parametrize("mode", [1, 2], ids=["8wave", "4wave"])
@pytest.mark.parametrize("case", P2_CASES, ids=[c[0] for c in P2_CASES])
def test_pipelined_p2_kernels_bit_exact(pai, case, mode):
    """gg_p2.hip (off by default, tunable fwd_p2): the one-workgroup-per-CU pipelined variants give the same bits."""
    used = _run_case(pai, case, tunables=(("fwd_p2", mode),))
    assert any(u.startswith("gg_fwd_p2_k<") for u in used[:2]), used


@pytest.mark.parametrize("case", P2_CASES, ids=[c[0] for c in P2_CASES])
def test_mfma_32x32x16_form_bit_exact(pai, case):
    """gg_fwd_patch32_k (off by default, tunable fwd_m32): the v_mfma_f32_32x32x16_bf16 form of the patch-resident
    kernel -- other LDS image, other accumulator layout, other epilogue mapping -- gives the same bits."""
    used = _run_case(pai, case, tunables=(("fwd_m32", 1),))
    assert any(u.startswith("gg_fwd_patch32_k<") for u in used[:2]), used


# layers whose forward or input gradient gives gg_fwd_bd_k >= 512 workgroups; cfg2_D1 / cfg2_D2: grids of several rounds,
# where the two workgroups of a CU run out of phase (the setting that exposed a wait counting an LDS-DMA together with
# register loads, see gg_bd.hip)
BD_CASES = [c for c in CASES if c[0] in ("cfg2_enc2", "cfg2_dec4", "cfg2_dec5", "cfg2_D1", "cfg2_D2", "cfg2_D3")]


@pytest.mark.parametrize("mode", [1, 2], ids=["2x2", "wide"])
@pytest.mark.parametrize("case", BD_CASES, ids=[c[0] for c in BD_CASES])
def test_register_direct_weight_kernel_bit_exact(pai, case, mode):
    """gg_bd.hip: the patch-resident kernel that feeds the matrix cores their weights straight from a fragment-major
    copy of the filter pack (pai_pack_frag) -- other operand path, other tile split, accumulators in the accumulation
    registers -- gives the same bits; without the copy (pack_flags = 0) the default kernels run."""
    used = _run_case(pai, case, tunables=(("fwd_bd", mode),), frag=True)
    want = "gg_fwd_bd_k<true, " if mode == 2 else "gg_fwd_bd_k<false, "
    assert any(u.startswith(want) for u in used[:2]), used
    assert not any(u.startswith("gg_fwd_bd_k") for u in _run_case(pai, case, tunables=(("fwd_bd", mode),))[:2])


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
