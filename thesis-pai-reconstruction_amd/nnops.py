"""Op-level autograd bridges for networks that are composed freely from convolution blocks (the residual
U-Net family, reference models/res_unet.py): one ``torch.autograd.Function`` per fused block
Conv2d (k = 1 | 3, optionally grouped) -> BatchNorm2d -> activation, plus MaxPool2d(2), nearest Upsample(2),
the residual sum and Dropout2d.  PyTorch supplies the tape and the parameter plumbing; every number comes from
``libpai_hip.so``.  Tensors between ops are NHWC ``[N, H, W, C]`` in the storage dtype (fp32 or bf16).

This is the composable counterpart of the hand-scheduled engines (engine.py / attention.py): simpler and
slower (each block is conv -> finalize -> apply, backward is the two-pass BatchNorm form), used where the
topology is not fixed.  Grouped 3x3 convolutions (ResNeXt, groups = 32, 4 channels per group) run on the dense
MFMA kernels with block-diagonal filters: the 32x extra FLOPs are cheaper there than a vector-ALU kernel, the
expansion / extraction of the filter is host-side plumbing on a 147 K-element tensor.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH


def _check(x):
    if not x.is_cuda:
        raise ops.PaiError("pai nnops need HIP device tensors (no CPU fallback exists)")
    if x.dim() != 4 or not x.is_contiguous():
        raise ops.PaiError("pai nnops take contiguous NHWC tensors [N, H, W, C]")


def to_nhwc(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 NCHW image batch -> NHWC storage-dtype tensor."""
    n, c, h, w = x.shape
    xs = x.to(torch.float32)
    xs = xs.reshape(n, h, w, 1) if c == 1 else xs.permute(0, 2, 3, 1)
    xs = xs.contiguous()
    if dtype == torch.float32:
        return xs
    out = torch.empty(xs.shape, dtype=dtype, device=xs.device)
    ops.cast(xs, out)
    return out


def _dense_fwd_pack(weight: torch.Tensor, groups: int) -> torch.Tensor:
    """torch Conv2d weight [Cout, Cin/groups, kh, kw] -> fp32 fwd pack [Cout][kh][kw][Cin] (block-diagonal for
    groups > 1): one library launch (pai_filter_to_dense); a 1x1 filter without groups already IS that layout."""
    cout, cig, kh, kw = weight.shape
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.to(torch.float32).contiguous()
    if groups == 1 and kh * kw == 1:
        return w.view(cout, 1, 1, cig)
    dense = torch.empty(cout, kh, kw, cig * groups, dtype=torch.float32, device=weight.device)
    ops.filter_to_dense(w, cout, cig, kh * kw, groups, dense)
    return dense


def _grad_from_fwd_pack(dw: torch.Tensor, weight: torch.Tensor, groups: int) -> torch.Tensor:
    cout, cig, kh, kw = weight.shape
    if groups == 1 and kh * kw == 1:
        return dw.view(cout, cig, 1, 1)
    out = torch.empty(cout, cig, kh, kw, dtype=torch.float32, device=dw.device)
    ops.filter_grad_from_dense(dw, cout, cig, kh * kw, groups, out)
    return out


_PREPACK = {}       # id(weight) -> (w_fwd, w_dgrad, weakref, version): bf16 packs made by prepack() for the forward pass that follows; single use (take_prepacked)


def prepack(module: torch.nn.Module, dtype) -> None:
    """bf16 packs of EVERY dense pointwise convolution of ``module`` in one launch (``pai_pack_weights_multi``) at the start
    of a forward pass, instead of one ``pack64_k`` launch in front of each convolution (40 of the ~600 launches on the main
    stream of a residual U-Net step, each 5 us + a launch gap in the launch-bound coarse levels).  The packs live for this
    forward pass only -- ``ConvBNAct.forward`` takes its layer's pair out of the table -- so a weight changed by anything
    (optimizer, ``load_state_dict``, an EMA swap) is simply packed again by the next call."""
    _PREPACK.clear()
    if dtype != torch.bfloat16 or os.environ.get("PAI_NO_PREPACK", "0") not in ("", "0"):
        return
    convs = getattr(module, "_pai_pointwise", None)
    if convs is None:
        convs = [m for m in module.modules()
                 if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.groups == 1 and m.stride == (1, 1)
                 and m.in_channels % 64 == 0 and m.out_channels % 64 == 0]
        object.__setattr__(module, "_pai_pointwise", convs)
    convs = [c for c in convs if c.weight.dtype == torch.float32 and c.weight.is_contiguous() and c.weight.is_cuda]
    if not convs:
        return
    buf = torch.empty(2 * sum(c.weight.numel() for c in convs), dtype=dtype, device=convs[0].weight.device)
    items, off = [], 0
    for c in convs:
        n = c.weight.numel()
        wf, wd = buf[off:off + n], buf[off + n:off + 2 * n]
        off += 2 * n
        items.append((c.weight.detach(), c.out_channels, 1, c.in_channels, wf, wd))
        _PREPACK[id(c.weight)] = (wf, wd, weakref.ref(c.weight), c.weight._version)
    ops.pack_weights_multi(items)


def take_prepacked(weight):
    """The pair of bf16 packs ``prepack`` made for ``weight``, or None.  An entry is only good for the tensor it was made
    from, at the version it had then: one left behind by a forward pass that never reached its convolution (an exception, a
    conditional path) must not serve a later call after an optimizer step / EMA swap, nor a new tensor that got the id."""
    ent = _PREPACK.pop(id(weight), None)
    if ent is None:
        return None
    wf, wd, ref, version = ent
    if ref() is not weight or weight._version != version:
        return None
    return wf, wd


_ZEROS = {}


def _zero_grad(n: int, device) -> torch.Tensor:
    """The gradient of a conv bias in front of a BatchNorm: identically zero (the BatchNorm subtracts the batch mean).  A view
    of one per-device buffer of zeros that nothing ever writes -- no fill launch per layer and step (59 of them in a
    residual U-Net step).  In-place scaling of such a gradient (clipping, unscaling) leaves it what it is; code that ADDS
    into ``.grad`` of these biases in place would have to replace the tensor first."""
    key = str(device)
    buf = _ZEROS.get(key)
    if buf is None or buf.numel() < n:
        buf = _ZEROS[key] = torch.zeros(max(n, 4096), dtype=torch.float32, device=device)
    return buf[:n]


class _WgradStream:
    """The weight gradients of the composable networks leave the critical path of the backward pass the way the Pix2Pix
    engine's do (engine._SideStream): each is issued on ONE second stream per device, ordered after the launch that produced
    its dz, and runs beside the next layers' BatchNorm-backward passes (HBM-bound) and input gradients.  ``join`` -- queued as
    the end-of-backward callback of the autograd engine, and called again by ``manual_backward`` and the optimizers -- makes
    the stream the backward pass was started from wait for them.  One
    stream: the library's weight-gradient slab buffer is used by one launch at a time.  PAI_NO_OVERLAP=1 turns it off."""

    KEEP_BYTES = 48 << 30       # upper bound; per backward pass the budget is also half of the free device memory

    def __init__(self):
        import os
        self.on = os.environ.get("PAI_NO_OVERLAP", "0") in ("", "0")
        self.streams = {}
        self.pending = set()
        self.keep, self.kept_bytes = [], 0
        self.budget = self.KEEP_BYTES
        self.seen = set()           # id() of the parameters whose gradient this backward pass has already produced
        self.queued = False         # the join is queued as this backward pass's end-of-pass callback

    @staticmethod
    def _hooked(p) -> bool:
        return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))

    def run(self, device, reads, fn, params=()):
        """fn() -> (results, temporaries) with the library's launches on the side stream.  `reads`: the tensors those
        launches read; they and the temporaries were allocated on the main stream and must outlive the side stream's use.
        A parameter that already HAS a gradient makes autograd add the new one to it on the main stream, during the backward
        pass: such a layer's weight gradient stays on the main stream.  So does that of a parameter used by a SECOND node
        of this graph (autograd sums the two gradients on the main stream before the join) and of one with tensor /
        post-accumulate hooks (they read the gradient the moment it is returned)."""
        if not self.on:
            return fn()[0]
        if not self.queued:
            # the backward pass ends with the join: callers of a bare loss.backward() read complete gradients as well, and
            # the per-pass bookkeeping (`seen`) is reset there
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.join)
                self.queued = True
            except RuntimeError:
                self.seen.clear()   # not inside an engine-driven backward pass (a Function's backward called by hand): the explicit joins remain
        live = [p for p in params if p is not None]
        again = any(id(p) in self.seen for p in live)
        self.seen.update(id(p) for p in live)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if again or any(p.grad is not None or self._hooked(p) for p in live):
            # main-stream fallback.  Weight gradients of this pass may still be running on the side stream: autograd adds
            # the first use's gradient of a re-used parameter to this one ON THE MAIN STREAM before the end-of-pass join,
            # a hook reads it the moment it is returned, and the library's slab workspace serves one launch at a time --
            # so the main stream first waits for what the side stream has been given.
            if idx in self.pending:
                ops.stream_wait_last(torch.cuda.current_stream(idx), self.streams[idx])
            return fn()[0]
        if not self.pending:
            try:
                self.budget = min(self.KEEP_BYTES, torch.cuda.mem_get_info(idx)[0] // 2)
            except RuntimeError:
                self.budget = self.KEEP_BYTES
        s = self.streams.get(idx)
        if s is None:
            s = self.streams[idx] = torch.cuda.Stream(device=idx)
        ops.stream_wait_last(s, torch.cuda.current_stream(idx))
        with ops.on_stream(s):          # library launches only: fn allocates on the main stream, see `reads`
            out, temps = fn()
        reads = tuple(reads) + tuple(temps)
        # main-stream tensors the side stream reads stay referenced until the join (the main stream then waits for the side
        # stream before anything can reuse their memory) -- record_stream costs an allocator event per tensor, ~8 ms of host
        # time per residual U-Net step; past KEEP_BYTES it is used after all
        nbytes = sum(t.numel() * t.element_size() for t in reads if t is not None)
        if self.kept_bytes + nbytes <= self.budget:
            self.keep.append(reads)
            self.kept_bytes += nbytes
        else:
            for t in reads:
                if t is not None:
                    t.record_stream(s)
        self.pending.add(idx)
        return out

    def join(self):
        for idx in tuple(self.pending):
            ops.stream_wait_last(torch.cuda.current_stream(idx), self.streams[idx])
        self.pending.clear()
        self.keep, self.kept_bytes = [], 0
        self.seen.clear()
        self.queued = False


WGRAD = _WgradStream()


def join_wgrads() -> None:
    """The current stream waits for every weight gradient issued on the side stream (no-op when none is pending)."""
    WGRAD.join()


class ConvBNAct(torch.autograd.Function):
    """act(BatchNorm2d(Conv2d(x))) with k = 1 or 3 ("same"), optional groups, optional norm.

    Replaces the aten::convolution / native_batch_norm / relu (+ backward) calls behind the ``nn.Sequential``
    blocks of reference models/res_unet.py:58-64,86-95,147-163,66-69 and the bare convolutions at :265,308."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, gamma, beta, bn, training, n_updates, act, groups, dtype, out_f32, defer=None,
                pre=None, pre_gamma=None, pre_beta=None, pre_act=ACT_NONE):
        # x2 (optional): a second NHWC tensor read as if concatenated behind x along C -- the torch.cat in front of the
        # decoder blocks (reference models/res_unet.py:327, models/trans_unet.py:113) never materialises
        # pre (optional): x is the RAW output of the producing convolution and `pre` the holder of its BatchNorm (a
        # ``defer`` dict): this layer and its weight gradient read pre_act(BN(x)) on load (``ops.conv_fwd_pro``), the activated
        # tensor is never written, and the producer's BatchNorm backward happens HERE (pre_gamma / pre_beta get its gradients)
        _check(x)
        N, H, W, C1 = x.shape
        C2 = 0
        if x2 is not None:
            _check(x2)
            if x2.shape[:3] != x.shape[:3] or x2.dtype != x.dtype:
                raise ops.PaiError("ConvBNAct: the two sources differ in shape or dtype")
            C2 = x2.shape[3]
        Cin = C1 + C2
        Cout, _, k, _ = weight.shape
        # grouped 3x3 (ResNeXt): block-diagonal dense packs + the groups hint (16-channel slices skip the zero blocks)
        hint = _groups_hint(groups, k, Cin, Cout)
        d = ops.make_desc(dtype, 0, N, H, W, C1, C2, Cout, 1, 0, 0, act if bn is None else ACT_NONE, kernel=k,
                          groups=hint)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), x.device)
        if Cin <= 2 or Cout <= 2:            # thin layers (in_conv / out): skinny-GEMM scratch of gg_thin.hip
            ops.ensure_scratch(ops.scratch_bytes_for([d]), x.device)
        # weight-gradient workspace: pixel-split slabs of the patch kernels, partial blocks of the grouped 3 x 3 gradient
        ops.ensure_wgrad_workspace([d], x.device)
        packed = take_prepacked(weight) if (k == 1 and groups == 1 and dtype == torch.bfloat16) else None
        wm = _dense_fwd_pack(weight, groups)
        if packed is not None:
            wf, wd = packed
        elif dtype == torch.float32:
            wf = wm
            wd = torch.empty_like(wm)
            ops.pack_weights(dtype, wm, Cout, k * k, Cin, None, wd)
        else:
            wf = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            wd = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            ops.pack_weights(dtype, wm, Cout, k * k, Cin, wf, wd)
        M = N * H * W
        b32 = None if bias is None else bias.detach().float()
        ctx.d, ctx.act, ctx.groups, ctx.dtype, ctx.has_bn, ctx.out_f32 = d, act, groups, dtype, bn is not None, out_f32
        ctx.has_x2 = x2 is not None
        ctx.bias_ref = bias
        ctx.deferred = False
        ctx.pre_act = pre_act if pre is not None else None
        extra = [x2] if x2 is not None else []
        if pre is not None:
            if x2 is not None or out_f32 or (bn is None and act != ACT_NONE) or not ops.conv_prologue_ok(d):
                raise ops.PaiError("ConvBNAct: this layer cannot read its input through a prologue")
            ctx.pre_training = pre["training"]
            extra = [pre["scale"], pre["shift"], pre["mean"], pre["rstd"], pre_gamma]

        def conv(**kw):        # the convolution proper: raw output (+ statistics) through the prologue, or the plain call
            if pre is None:
                return ops.conv_fwd(d, x, x2, wf, b32, **kw)
            return ops.conv_fwd_pro(d, x, wf, b32, kw["y_raw"], kw.get("stats"), pre["scale"], pre["shift"], pre_act)

        if bn is None:
            if out_f32:                       # final conv + tanh (reference :307-315): fp32 NCHW-compatible output
                out = torch.empty(N, H, W, Cout, dtype=torch.float32, device=x.device)
                ops.conv_fwd(d, x, x2, wf, b32, y_f32=out)
            elif act == ACT_NONE:
                out = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
                conv(y_raw=out)
            else:
                out = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
                ops.conv_fwd(d, x, x2, wf, b32, y_act=out)
            ctx.save_for_backward(x, out, wd, weight, *extra)
            return out
        z = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
        f32 = dict(dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        scale, shift = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        if training:
            rows = ops.conv_fwd_stats_rows(d)
            stats = torch.empty(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, **f32)
            conv(y_raw=z, stats=stats)
            mom = bn.momentum if bn.momentum is not None else 0.1
            ops.bn_finalize(stats, rows, Cout, M, gamma.detach(), beta.detach(), float(bn.eps), float(mom), n_updates,
                            bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, rstd, scale, shift)
        else:
            conv(y_raw=z)
            ops.bn_eval_coeffs(Cout, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, float(bn.eps),
                               scale, shift)
        if defer is not None:
            # the normalisation itself belongs to the caller's BNTail (residual sum): z goes out raw, its batch statistics in
            # `defer`; the gradient that comes back is dz and gamma / beta are not this node's inputs
            if act != ACT_NONE:
                raise ops.PaiError("ConvBNAct: a deferred BatchNorm carries no activation of its own")
            defer.update(mean=mean, rstd=rstd, scale=scale, shift=shift, training=training)
            ctx.deferred = True
            ctx.save_for_backward(x, z, wd, weight, *extra)
            return z
        out = torch.empty_like(z)
        ops.bn_apply(dtype, z, M, Cout, scale, shift, act, out)
        ctx.training = training
        ctx.save_for_backward(x, out, wd, weight, z, mean, rstd, gamma, scale, shift, *extra)
        return out

    @staticmethod
    def backward(ctx, g):
        d, act, dtype = ctx.d, ctx.act, ctx.dtype
        N, H, W, C1, C2, Cout = d.N, d.H, d.W, d.C1, d.C2, d.Cout
        Cin = C1 + C2
        x2 = ctx.saved_tensors[-1] if ctx.has_x2 else None
        M = N * H * W
        g = g.contiguous()
        dev = g.device
        f32 = dict(dtype=torch.float32, device=dev)
        dgamma = dbeta = None
        if ctx.deferred:
            x, _, wd, weight = ctx.saved_tensors[:4]
            dz = g
        elif not ctx.has_bn:
            x, out, wd, weight = ctx.saved_tensors[:4]
            dz = torch.empty(N, H, W, Cout, dtype=dtype, device=dev)
            if ctx.out_f32:
                ops.tanh_bwd(dtype, out, g.float(), None, dz) if act == ACT_TANH else ops.cast(g.float(), dz)
            elif act == ACT_NONE:
                dz = g
            else:
                ops.act_bwd(dtype, g, act, None, ACT_NONE, out, g.numel(), dz)
        else:
            x, out, wd, weight, z, mean, rstd, gamma, scale, shift = ctx.saved_tensors[:10]
            if not ctx.training:
                raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
            part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * Cout, **f32)
            sums = torch.empty(2 * Cout, **f32)
            # sums = [sum du | sum du * xhat] IS [dbeta | dgamma]: nothing is accumulated into zero-filled tensors (two fill
            # launches per BatchNorm: ~0.4 ms of a ResNeXt step in 3-us kernels)
            dbeta, dgamma = sums[:Cout], sums[Cout:]
            dz = torch.empty_like(z)
            # five tensor passes: pass 1 reads g and z and stores nothing, pass 2 reads them again and writes dz
            if act != ACT_NONE:     # the activation's sign from z * scale + shift: `out` is not read again, du never stored
                ops.bn_bwd_reduce_affine(dtype, g, act, None, ACT_NONE, z, M, Cout, scale, shift, mean, rstd, None, part, sums,
                                         None, None)
                ops.bn_bwd_apply_affine(dtype, g, act, z, M, Cout, scale, shift, mean, rstd, gamma.detach(), sums, dz)
            else:                   # du IS g
                ops.bn_bwd_reduce(dtype, g, act, None, ACT_NONE, None, z, M, Cout, mean, rstd, None, part, sums, None, None)
                ops.bn_bwd_apply(dtype, g, z, M, Cout, mean, rstd, gamma.detach(), sums, dz)
        k = weight.shape[2]
        # a conv bias in front of a BatchNorm has an identically zero gradient
        with_bias = ctx.needs_input_grad[3] and not ctx.has_bn       # inputs: x, x2, weight, bias, gamma, beta, ...
        groups = ctx.groups
        pre = ctx.pre_act is not None
        if pre:
            psc, psh, pmean, prstd, pgam = ctx.saved_tensors[-5:]

        def wgrad():
            dw = torch.empty(Cout * k * k * Cin, **f32)
            dbias = None
            if ctx.needs_input_grad[3]:
                dbias = torch.empty(Cout, **f32) if with_bias else _zero_grad(Cout, dev)
            if pre:         # x is the producer's raw output: its BatchNorm + activation on load, as in the forward pass
                ops.conv_wgrad_pro(d, x, dz, dw, dbias if with_bias else None, True, psc, psh, ctx.pre_act)
            else:
                ops.conv_wgrad_overwrite(d, x, x2, dz, dw, dbias if with_bias else None)
            return (_grad_from_fwd_pack(dw, weight, groups), dbias), (dw,)

        gw, dbias = WGRAD.run(dev, (x, x2, dz) + ((psc, psh) if pre else ()), wgrad, (weight, ctx.bias_ref))
        dx = dx2 = dpg = dpb = None
        fused_bwd = pre and ctx.needs_input_grad[0] and ctx.pre_training and fuse_dgrad_bn() \
            and ops.conv_kernel_name(d, 1).startswith("pwx_k")
        if fused_bwd:
            # the streaming pointwise kernel forms du = act'(x * scale + shift) * dgrad and the BatchNorm partial sums in its
            # store (pai_conv_dgrad_bn): pass 1 of the producer's BatchNorm backward costs one read of x instead of a read of
            # dx and x; pass 2 turns du into dz in place
            part = torch.empty(ops.conv_dgrad_bn_rows_max(d) * 2 * C1, **f32)
            sums = torch.empty(2 * C1, **f32)
            dx = torch.empty(N, H, W, C1, dtype=dtype, device=dev)
            rows = ops.conv_dgrad_bn(d, dz, wd, dx, None, x, ctx.pre_act, scale=psc, shift=psh, mean=pmean, rstd=prstd,
                                     partials=part)
            ops.bn_bwd_finalize(part, rows, C1, sums, None, None)
            ops.bn_bwd_apply(dtype, dx, x, M, C1, pmean, prstd, pgam.detach(), sums, dx)
            dpb, dpg = sums[:C1], sums[C1:]
        elif ctx.needs_input_grad[0] or (ctx.has_x2 and ctx.needs_input_grad[1]):
            dx = torch.empty(N, H, W, C1, dtype=dtype, device=dev)
            dx2 = torch.empty(N, H, W, C2, dtype=dtype, device=dev) if ctx.has_x2 else None
            ops.conv_dgrad(d, dz, wd, dx, dx2)
        if pre and dx is not None and not fused_bwd:
            # dx is the gradient behind the producer's BatchNorm + activation: its backward (two passes over dx and x = z),
            # the producing convolution gets dz and the BatchNorm parameters their gradients from here
            if not ctx.pre_training:
                raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
            part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C1, **f32)
            sums = torch.empty(2 * C1, **f32)
            dzp = torch.empty_like(dx)
            if ctx.pre_act != ACT_NONE:
                ops.bn_bwd_reduce_affine(dtype, dx, ctx.pre_act, None, ACT_NONE, x, M, C1, psc, psh, pmean, prstd, None, part,
                                         sums, None, None)
                ops.bn_bwd_apply_affine(dtype, dx, ctx.pre_act, x, M, C1, psc, psh, pmean, prstd, pgam.detach(), sums, dzp)
            else:
                ops.bn_bwd_reduce(dtype, dx, ACT_NONE, None, ACT_NONE, None, x, M, C1, pmean, prstd, None, part, sums, None, None)
                ops.bn_bwd_apply(dtype, dx, x, M, C1, pmean, prstd, pgam.detach(), sums, dzp)
            dx, dpb, dpg = dzp, sums[:C1], sums[C1:]
        return dx, dx2, gw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, dpg, dpb, None


class ConvK4S2(torch.autograd.Function):
    """act(nn.Conv2d(k4, s2, p1)(x)) on an NHWC tensor: the convolution of DiscriminatorBlock as a stand-alone op (reference
    models/wrapper.py:196-203).  The networks of this package run these layers inside their engines; this Function exists
    for a DiscriminatorBlock used on its own (``norm=True`` included, which the reference's Discriminator never builds)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, dtype):
        _check(x)
        N, H, W, Cin = x.shape
        Cout = weight.shape[0]
        if H % 2 or W % 2 or tuple(weight.shape[1:]) != (Cin, 4, 4):
            raise ops.PaiError("ConvK4S2: even image sizes and a [Cout, Cin, 4, 4] filter")
        d = ops.make_desc(dtype, 0, N, H, W, Cin, 0, Cout, 2, 0, 0, act)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), x.device)
        ops.ensure_scratch(ops.scratch_bytes_for([d]), x.device)
        ops.ensure_wgrad_workspace([d], x.device)
        wm = _dense_fwd_pack(weight, 1)
        if dtype == torch.float32:
            wf, wd = wm, torch.empty_like(wm)
            ops.pack_weights(dtype, wm, Cout, 16, Cin, None, wd)
        else:
            wf = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            wd = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            ops.pack_weights(dtype, wm, Cout, 16, Cin, wf, wd)
        out = torch.empty(N, H // 2, W // 2, Cout, dtype=dtype, device=x.device)
        b32 = None if bias is None else bias.detach().float()
        if act == ACT_NONE:
            ops.conv_fwd(d, x, None, wf, b32, y_raw=out)
        else:
            ops.conv_fwd(d, x, None, wf, b32, y_act=out)
        ctx.d, ctx.act, ctx.dtype = d, act, dtype
        ctx.save_for_backward(x, out, wd, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, wd, weight = ctx.saved_tensors
        d, act, dtype = ctx.d, ctx.act, ctx.dtype
        g = g.contiguous()
        dz = g
        if act != ACT_NONE:
            dz = torch.empty_like(out)
            ops.act_bwd(dtype, g, act, None, ACT_NONE, out, g.numel(), dz)
        f32 = dict(dtype=torch.float32, device=g.device)
        dw = torch.empty(weight.numel(), **f32)
        dbias = torch.empty(d.Cout, **f32) if ctx.needs_input_grad[2] else None
        ops.conv_wgrad_overwrite(d, x, None, dz, dw, dbias)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.conv_dgrad(d, dz, wd, dx, None)
        return dx, _grad_from_fwd_pack(dw, weight, 1), dbias, None, None


class InstanceNormAct(torch.autograd.Function):
    """act(nn.InstanceNorm2d(C)(x)) on an NHWC tensor (affine=False, no running statistics: the module DiscriminatorBlock
    builds for ``norm=True``, reference models/wrapper.py:203)."""

    @staticmethod
    def forward(ctx, x, eps, act):
        _check(x)
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N * C, dtype=torch.float32, device=x.device)
        ops.instnorm_fwd(x.dtype, x, N, H * W, C, eps, act, y, mean, rstd)
        ctx.act = act
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd = ctx.saved_tensors
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        ops.instnorm_bwd(x.dtype, g.contiguous(), x, N, H * W, C, ctx.act, mean, rstd, dx)
        return dx, None, None


class ToStorage(torch.autograd.Function):
    """fp32 NHWC -> storage dtype (``pai_cast``) with the gradient cast back: the differentiable form of ``to_nhwc``'s last
    step, for stand-alone blocks whose input needs a gradient."""

    @staticmethod
    def forward(ctx, x, dtype):
        if dtype == torch.float32:
            return x
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
        ops.cast(x.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g.float(), None


class MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2) (reference models/res_unet.py:199)."""

    @staticmethod
    def forward(ctx, x):
        _check(x)
        N, H, W, C = x.shape
        out = torch.empty(N, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
        idx = torch.empty(out.numel(), dtype=torch.uint8, device=x.device)
        ops.maxpool2(x.dtype, x, N, H, W, C, out, idx)
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=g.dtype, device=g.device)
        ops.maxpool2_bwd(g.dtype, g.contiguous(), idx, N, H, W, C, dx)
        return dx


class Upsample2(torch.autograd.Function):
    """nn.Upsample(scale_factor=2), mode 'nearest' (reference models/res_unet.py:231)."""

    @staticmethod
    def forward(ctx, x):
        _check(x)
        N, H, W, C = x.shape
        out = torch.empty(N, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
        ops.upsample2(x.dtype, x, N, H, W, C, out)
        ctx.shape = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=g.dtype, device=g.device)
        ops.upsample2_bwd(g.dtype, g.contiguous(), N, H, W, C, dx)
        return dx


class AddAct(torch.autograd.Function):
    """act(a + b): the residual sum (reference models/res_unet.py:74,105,171)."""

    @staticmethod
    def forward(ctx, a, b, act):
        _check(a)
        _check(b)
        out = torch.empty_like(a)
        ops.add_act(a.dtype, a, b, act, out)
        ctx.act = act
        if act != ACT_NONE:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.act == ACT_NONE:
            return g, g, None
        (out,) = ctx.saved_tensors
        d = torch.empty_like(out)
        ops.act_bwd(out.dtype, g.contiguous(), ctx.act, None, ACT_NONE, out, out.numel(), d)
        return d, d, None


class BNTail(torch.autograd.Function):
    """act(act_a(BN_a(za)) + BN_b(zb)) -- the tail of a residual block (reference models/res_unet.py:74,105,160-171,
    models/trans_unet.py:227-236): the BatchNorm (+ ReLU: ResNeXt) of the residual branch, the BatchNorm of the skip branch (``hb`` None: the
    skip is the identity and ``zb`` is added as it is), the sum and the activation behind it in ONE pass over three tensors
    (``pai_bn2_add_act``) instead of bn_apply + bn_apply + add_act over seven.  ``za`` / ``zb`` come out of
    ``ConvBNAct(..., defer=ha / hb)``: raw convolution outputs with their batch statistics in the holder."""

    @staticmethod
    def forward(ctx, za, gamma_a, beta_a, ha, zb, gamma_b, beta_b, hb, act_a, act):
        _check(za)
        _check(zb)
        if za.shape != zb.shape or za.dtype != zb.dtype:
            raise ops.PaiError("BNTail: the two branches differ in shape or dtype")
        N, H, W, Cc = za.shape
        out = torch.empty_like(za)
        ops.bn2_add_act(za.dtype, za, ha["scale"], ha["shift"], zb, hb["scale"] if hb else None, hb["shift"] if hb else None,
                        N * H * W, Cc, act_a, act, out)
        ctx.act_a, ctx.act, ctx.ha, ctx.hb = act_a, act, ha, hb
        ctx.save_for_backward(za, zb, out if act != ACT_NONE else None, gamma_a, gamma_b)
        return out

    @staticmethod
    def backward(ctx, g):
        za, zb, out, gamma_a, gamma_b = ctx.saved_tensors
        N, H, W, Cc = za.shape
        M, dtype = N * H * W, za.dtype
        f32 = dict(dtype=torch.float32, device=za.device)
        g = g.contiguous()
        if ctx.act != ACT_NONE:
            d = torch.empty_like(out)
            ops.act_bwd(dtype, g, ctx.act, None, ACT_NONE, out, out.numel(), d)
        else:
            d = g

        def bn_bwd(z, h, gamma, act):
            if not h["training"]:
                raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
            part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * Cc, **f32)
            sums = torch.empty(2 * Cc, **f32)
            dz = torch.empty_like(z)
            if act != ACT_NONE:     # the branch's own activation: its sign from z * scale + shift, as in ConvBNAct.backward
                ops.bn_bwd_reduce_affine(dtype, d, act, None, ACT_NONE, z, M, Cc, h["scale"], h["shift"], h["mean"], h["rstd"],
                                         None, part, sums, None, None)
                ops.bn_bwd_apply_affine(dtype, d, act, z, M, Cc, h["scale"], h["shift"], h["mean"], h["rstd"], gamma.detach(),
                                        sums, dz)
            else:
                ops.bn_bwd_reduce(dtype, d, ACT_NONE, None, ACT_NONE, None, z, M, Cc, h["mean"], h["rstd"], None, part, sums,
                                  None, None)
                ops.bn_bwd_apply(dtype, d, z, M, Cc, h["mean"], h["rstd"], gamma.detach(), sums, dz)
            return dz, sums[Cc:], sums[:Cc]

        if ctx.hb is not None and ctx.act_a in (ACT_NONE, ACT_RELU, ACT_LRELU):
            # both BatchNorms read the same gradient d: one pass over (d, za, zb) for the two sets of sums, one for the two dz
            # (pai_bn2_bwd_*; 3 + 5 tensor passes instead of 4 + 6)
            ha, hb = ctx.ha, ctx.hb
            if not (ha["training"] and hb["training"]):
                raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
            rows = ops.bn_bwd_partial_rows(M)
            part_a, part_b = torch.empty(rows * 2 * Cc, **f32), torch.empty(rows * 2 * Cc, **f32)
            sums_a, sums_b = torch.empty(2 * Cc, **f32), torch.empty(2 * Cc, **f32)
            dza, dzb = torch.empty_like(za), torch.empty_like(zb)
            ops.bn2_bwd_reduce(dtype, d, ctx.act_a, za, zb, M, Cc, ha["scale"], ha["shift"], ha["mean"], ha["rstd"], hb["mean"],
                               hb["rstd"], part_a, part_b, sums_a, sums_b)
            ops.bn2_bwd_apply(dtype, d, ctx.act_a, za, zb, M, Cc, ha["scale"], ha["shift"], ha["mean"], ha["rstd"],
                              gamma_a.detach(), sums_a, hb["mean"], hb["rstd"], gamma_b.detach(), sums_b, dza, dzb)
            dga, dba, dgb, dbb = sums_a[Cc:], sums_a[:Cc], sums_b[Cc:], sums_b[:Cc]
        else:
            dza, dga, dba = bn_bwd(za, ctx.ha, gamma_a, ctx.act_a)
            if ctx.hb is not None:
                dzb, dgb, dbb = bn_bwd(zb, ctx.hb, gamma_b, ACT_NONE)
            else:
                dzb, dgb, dbb = d, None, None
        return dza, dga, dba, None, dzb, dgb, dbb, None, None, None


def fuse_tail() -> bool:
    """PAI_NO_BN_TAIL=1: the residual blocks' tails as separate bn_apply / bn_apply / add_act passes (A/B switch)."""
    import os
    return os.environ.get("PAI_NO_BN_TAIL", "0") in ("", "0")


def bn_tail(h, conv, bn, act_a, xs, skip_conv, skip_bn, act, training, n_updates, dtype, pre=None):
    """act(act_a(BN(conv(h))) + BN_skip(conv_skip(xs))) (``skip_conv`` None: + xs) through ``BNTail``."""
    ha = {}
    za = conv_bn_act(h, conv, bn, ACT_NONE, training, n_updates, dtype, defer=ha, pre=pre)
    if skip_conv is None:
        return BNTail.apply(za, bn.weight, bn.bias, ha, xs, None, None, None, act_a, act)
    hb = {}
    zb = conv_bn_act(xs, skip_conv, skip_bn, ACT_NONE, training, n_updates, dtype, defer=hb)
    return BNTail.apply(za, bn.weight, bn.bias, ha, zb, skip_bn.weight, skip_bn.bias, hb, act_a, act)


class Dropout2d(torch.autograd.Function):
    """nn.Dropout2d in training mode with a caller-supplied [N, C] mask of {0, 1/(1-p)} (models/res_unet.py:230)."""

    @staticmethod
    def forward(ctx, x, mask):
        _check(x)
        N, H, W, C = x.shape
        out = torch.empty_like(x)
        ops.dropout2d(x.dtype, x, mask, N, H * W, C, out)
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        N, H, W, C = g.shape
        out = torch.empty_like(g)
        ops.dropout2d(g.dtype, g.contiguous(), mask, N, H * W, C, out)
        return out, None


class Fork(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (block input -> residual branch + skip branch, encoder feature ->
    next level + decoder, token stream -> sublayer + its residual; reference models/res_unet.py:165-171,
    models/trans_unet.py:227-236, nn.TransformerEncoderLayer).  Autograd would sum the two gradients with an aten add of
    its own; here the sum is a library launch (``pai_add_act``), so a whole step consists of launches a plan can own."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g2 if g1 is None else g1
        g1, g2 = g1.contiguous(), g2.contiguous()
        out = torch.empty_like(g1)
        ops.add_act(g1.dtype, g1, g2, ACT_NONE, out)
        return out


def fork(x):
    """Two handles of `x` (a tensor, or the pair a decoder block reads as a concatenation) for its two consumers."""
    if isinstance(x, tuple):
        a, b = zip(*(Fork.apply(t) for t in x))
        return tuple(a), tuple(b)
    return Fork.apply(x)


class SwapMid(torch.autograd.Function):
    """[A][B][C][D] -> [A][C][B][D] (``pai_swap_mid``): the patch rearrangement around the ViT bottleneck."""

    @staticmethod
    def forward(ctx, x, A, B, Cc, D):
        ctx.dims = (A, B, Cc, D)
        out = torch.empty_like(x)
        ops.swap_mid(x, A, B, Cc, D, out)
        return out

    @staticmethod
    def backward(ctx, g):
        A, B, Cc, D = ctx.dims
        g = g.contiguous()
        dx = torch.empty_like(g)
        ops.swap_mid(g, A, Cc, B, D, dx)
        return dx, None, None, None, None


def fuse_dgrad_bn() -> bool:
    """PAI_NO_DGRAD_BN=1: the producer's BatchNorm backward always as two passes behind a plain input gradient (A/B switch)."""
    return os.environ.get("PAI_NO_DGRAD_BN", "0") in ("", "0")


def no_prologue() -> bool:
    """PAI_NO_PROLOGUE=1: every BatchNorm + activation as a pass of its own (A/B switch)."""
    import os
    return os.environ.get("PAI_NO_PROLOGUE", "0") not in ("", "0")


def _groups_hint(groups, k, cin, cout) -> int:
    """pai_conv_desc.groups for a grouped 3 x 3 layer whose groups tile into 16-channel slices (ResNeXt), else 1 (dense)."""
    return groups if (groups > 1 and k == 3 and cin == cout and cin % 16 == 0 and 16 % (cin // groups) == 0) else 1


def can_prologue(x, nxt, dtype, act) -> bool:
    """Can convolution ``nxt`` read the NHWC tensor ``x`` (the raw output of the layer in front of it) through that layer's
    BatchNorm + activation (``ops.conv_prologue_ok``: forward AND weight gradient)?"""
    if dtype != torch.bfloat16 or act not in (ACT_NONE, ACT_RELU) or no_prologue():
        return False
    if isinstance(x, tuple):        # (the producer's own input may be a pair; its output is one tensor of the same extent)
        x = x[0]
    if not isinstance(nxt, torch.nn.Conv2d) or nxt.stride != (1, 1) or nxt.kernel_size not in ((1, 1), (3, 3)):
        return False
    k = nxt.kernel_size[0]
    hint = _groups_hint(nxt.groups, k, nxt.in_channels, nxt.out_channels)
    if nxt.padding != (k // 2, k // 2) or (nxt.groups > 1 and hint == 1):
        return False
    N, H, W, _ = x.shape
    d = ops.make_desc(dtype, 0, N, H, W, nxt.in_channels, 0, nxt.out_channels, 1, 0, 0, ACT_NONE, kernel=k, groups=hint)
    ops.ensure_wgrad_workspace([d], x.device)       # (the grouped weight gradient needs its workspace to say yes)
    return ops.conv_prologue_ok(d)


def conv_bn_act(x, conv, bn, act, training, n_updates, dtype, out_f32=False, defer=None, pre=None):
    """Run an ``nn.Conv2d`` (+ ``nn.BatchNorm2d``) parameter container through ConvBNAct.  ``x``: an NHWC tensor, or a pair
    ``(x1, x2)`` read as ``torch.cat([x1, x2], dim=3)`` without the concatenation being built.  ``defer`` (a dict): the
    BatchNorm is only measured (batch statistics, running averages), its normalisation is left to ``BNTail``."""
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    if defer is not None:
        if bn is None:
            raise ops.PaiError("conv_bn_act: defer needs a BatchNorm")
        gamma, beta = gamma.detach(), beta.detach()       # their gradients come out of BNTail
    x1, x2 = x if isinstance(x, tuple) else (x, None)
    if pre is not None:         # (holder, BatchNorm2d, activation) of the producing layer, see ConvBNAct.forward
        hold, pbn, pact = pre
        return ConvBNAct.apply(x1, x2, conv.weight, conv.bias, gamma, beta, bn, training, n_updates, act, conv.groups, dtype,
                               out_f32, defer, hold, pbn.weight, pbn.bias, pact)
    return ConvBNAct.apply(x1, x2, conv.weight, conv.bias, gamma, beta, bn, training, n_updates, act, conv.groups, dtype,
                           out_f32, defer)


def as_tensor(x):
    """The concatenation itself, for consumers that need one tensor (identity skips, standalone BatchNorm)."""
    return torch.cat(list(x), dim=3) if isinstance(x, tuple) else x


# ---- TransUNet (reference models/trans_unet.py) ----------------------------------------------------------------
class Subsample2(torch.autograd.Function):
    """x[:, ::2, ::2, :] of an NHWC tensor.  Conv2d(k3, s2, p1) = Subsample2(Conv2d(k3, s1, p1)) and
    Conv2d(k1, s2) = Conv2d(k1)(Subsample2(x)): the strided convolutions of the TransUNet EncoderBlock
    (reference models/trans_unet.py:203-227) on the stride-1 kernels."""

    @staticmethod
    def forward(ctx, x):
        _check(x)
        N, H, W, C = x.shape
        out = torch.empty(N, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
        ops.subsample2(x.dtype, x, N, H, W, C, out)
        ctx.shape = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=g.dtype, device=g.device)
        ops.subsample2_bwd(g.dtype, g.contiguous(), N, H, W, C, dx)
        return dx


class BNAct(torch.autograd.Function):
    """act(BatchNorm2d(z)) of a stored NHWC tensor (the BatchNorm behind a subsampled convolution, reference
    models/trans_unet.py:213-214)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, training, n_updates, act):
        _check(z)
        N, H, W, C = z.shape
        M, dtype = N * H * W, z.dtype
        f32 = dict(dtype=torch.float32, device=z.device)
        mean, rstd = torch.empty(C, **f32), torch.empty(C, **f32)
        scale, shift = torch.empty(C, **f32), torch.empty(C, **f32)
        if training:
            rows = ops.bn_stats_rows(M)
            stats = torch.empty(ops.bn_stats_buffer_rows(rows) * 2 * C, **f32)
            ops.bn_stats(dtype, z, M, C, stats)
            mom = bn.momentum if bn.momentum is not None else 0.1
            ops.bn_finalize(stats, rows, C, M, gamma.detach(), beta.detach(), float(bn.eps), float(mom), n_updates,
                            bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, rstd, scale, shift)
        else:
            ops.bn_eval_coeffs(C, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, float(bn.eps), scale, shift)
        out = torch.empty_like(z)
        ops.bn_apply(dtype, z, M, C, scale, shift, act, out)
        ctx.act, ctx.training = act, training
        ctx.save_for_backward(z, mean, rstd, gamma, scale, shift)
        return out

    @staticmethod
    def backward(ctx, g):
        z, mean, rstd, gamma, scale, shift = ctx.saved_tensors
        if not ctx.training:
            raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
        N, H, W, C = z.shape
        M, dtype, act = N * H * W, z.dtype, ctx.act
        f32 = dict(dtype=torch.float32, device=z.device)
        g = g.contiguous()
        part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C, **f32)
        sums = torch.empty(2 * C, **f32)
        dbeta, dgamma = sums[:C], sums[C:]      # see ConvBNAct.backward
        dz = torch.empty_like(z)
        if act != ACT_NONE:         # du is never stored (see ConvBNAct.backward)
            ops.bn_bwd_reduce_affine(dtype, g, act, None, ACT_NONE, z, M, C, scale, shift, mean, rstd, None, part, sums, None,
                                     None)
            ops.bn_bwd_apply_affine(dtype, g, act, z, M, C, scale, shift, mean, rstd, gamma.detach(), sums, dz)
        else:
            ops.bn_bwd_reduce(dtype, g, act, None, ACT_NONE, None, z, M, C, mean, rstd, None, part, sums, None, None)
            ops.bn_bwd_apply(dtype, g, z, M, C, mean, rstd, gamma.detach(), sums, dz)
        return dz, dgamma, dbeta, None, None, None, None


def _check_tokens(x):
    if not x.is_cuda:
        raise ops.PaiError("pai nnops need HIP device tensors (no CPU fallback exists)")
    if x.dim() != 2 or not x.is_contiguous():
        raise ops.PaiError("pai token ops take contiguous [tokens, features] tensors")


class Linear(torch.autograd.Function):
    """nn.Linear over token rows [M, K] -> [M, out] as a pointwise convolution of a 1 x M image (the gather-GEMM
    kernels with one tap): patch embedding, attention in / out projections and the feed-forward layers of the ViT
    bottleneck (reference models/trans_unet.py:143,151-156)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _check_tokens(x)
        M, K = x.shape
        out_f = weight.shape[0]
        dtype = x.dtype
        d = ops.make_desc(dtype, 0, 1, 1, M, K, 0, out_f, 1, 0, 0, ACT_NONE, kernel=1)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), x.device)
        wm = weight.detach()
        if dtype == torch.float32:
            wf = wm
            wd = torch.empty_like(wm)
            ops.pack_weights(dtype, wm, out_f, 1, K, None, wd)
        else:
            wf = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            wd = torch.empty(wm.numel(), dtype=dtype, device=x.device)
            ops.pack_weights(dtype, wm, out_f, 1, K, wf, wd)
        y = torch.empty(M, out_f, dtype=dtype, device=x.device)
        ops.conv_fwd(d, x, None, wf, None if bias is None else bias.detach(), y_raw=y)
        ctx.d = d
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, wd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, wd = ctx.saved_tensors
        d = ctx.d
        g = g.contiguous()
        M, K, out_f = d.W, d.C1, d.Cout
        f32 = dict(dtype=torch.float32, device=g.device)
        has_bias = ctx.has_bias

        def wgrad():
            dw = torch.empty(out_f, K, **f32)
            db = torch.empty(out_f, **f32) if has_bias else None
            ops.conv_wgrad_overwrite(d, x, None, g, dw, db)
            return (dw, db), ()

        dw, db = WGRAD.run(g.device, (x, g), wgrad, ctx.params)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=g.dtype, device=g.device)
            ops.conv_dgrad(d, g, wd, dx, None)
        return dx, dw, db


class LayerNorm(torch.autograd.Function):
    """post + LayerNorm(x + res) over the last dimension of [M, D] (nn.LayerNorm, reference models/trans_unet.py:142,144
    and norm1 / norm2 of nn.TransformerEncoderLayer; ``post`` = the pos_embedding added at :172, period P rows)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, post):
        _check_tokens(x)
        M, D = x.shape
        dtype = x.dtype
        f32 = dict(dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        s = torch.empty_like(x) if res is not None else None
        mean, rstd = torch.empty(M, **f32), torch.empty(M, **f32)
        P = 0
        if post is not None:
            P = post.numel() // D
            if M % P:
                raise ops.PaiError(f"LayerNorm: {M} rows are not a multiple of the addend's period {P}")
        ops.layernorm_fwd(dtype, x, res, M, D, gamma.detach(), beta.detach(), eps, None if post is None else post.detach(),
                          P, s, y, mean, rstd)
        ctx.P, ctx.has_res, ctx.has_post = P, res is not None, post is not None
        ctx.post_shape = None if post is None else post.shape
        ctx.save_for_backward(s if res is not None else x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        xs, gamma, mean, rstd = ctx.saved_tensors
        M, D = xs.shape
        g = g.contiguous()
        f32 = dict(dtype=torch.float32, device=g.device)
        dx = torch.empty_like(xs)
        dgb = torch.empty(2, D, **f32)
        part = torch.empty(ops.layernorm_partial_rows(M) * 2 * D, **f32)
        ops.layernorm_bwd(xs.dtype, g, xs, M, D, gamma.detach(), mean, rstd, dx, dgb, part)
        dpost = None
        if ctx.has_post:
            dpost = torch.empty(ctx.P * D, **f32)
            ops.zero_multi([dpost])
            ops.colsum(g.dtype, g, M // ctx.P, ctx.P * D, dpost)
            dpost = dpost.view(ctx.post_shape)
        return dx, (dx if ctx.has_res else None), dgb[1], dgb[0], None, dpost


class GELU(torch.autograd.Function):
    """erf GELU (activation="gelu" of nn.TransformerEncoderLayer, reference models/trans_unet.py:155)."""

    @staticmethod
    def forward(ctx, z):
        _check_tokens(z)
        out = torch.empty_like(z)
        ops.gelu(z.dtype, z, out)
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        dz = torch.empty_like(z)
        ops.gelu_bwd(z.dtype, g.contiguous(), z, dz)
        return dz


class MHACore(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v per (batch entry, head) on packed projections qkv [S * B, 3 E], row = s * B + b
    (nn.MultiheadAttention with batch_first = False: S is the image batch)."""

    @staticmethod
    def forward(ctx, qkv, S, B, heads, mask=None):
        _check_tokens(qkv)
        E = qkv.shape[1] // 3
        hd = E // heads
        if qkv.shape[0] != S * B or E * 3 != qkv.shape[1] or hd * heads != E:
            raise ops.PaiError(f"MHACore: qkv {tuple(qkv.shape)} does not match S={S} B={B} heads={heads}")
        out = torch.empty(S * B, E, dtype=qkv.dtype, device=qkv.device)
        probs = torch.empty(B * heads * S * S, dtype=torch.float32, device=qkv.device)
        if mask is not None and (mask.numel() != probs.numel() or not mask.is_contiguous()):
            raise ops.PaiError("MHACore: the attention-dropout mask must be a contiguous fp32 [B * heads, S, S] tensor")
        ops.mha_fwd(qkv.dtype, qkv, S, B, heads, hd, out, probs, mask)
        ctx.dims = (S, B, heads, hd)
        ctx.has_mask = mask is not None
        ctx.save_for_backward(qkv, probs, *([mask] if mask is not None else []))
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, probs = ctx.saved_tensors[:2]
        mask = ctx.saved_tensors[2] if ctx.has_mask else None
        S, B, heads, hd = ctx.dims
        dqkv = torch.empty_like(qkv)
        ds = torch.empty_like(probs)
        ops.mha_bwd(qkv.dtype, g.contiguous(), qkv, probs, S, B, heads, hd, dqkv, ds, mask)
        return dqkv, None, None, None, None


class TokenDropout(torch.autograd.Function):
    """nn.Dropout on a token tensor [M, D] with a caller-supplied fp32 mask [M, D] of {0, 1 / (1 - p)} (dropout1 /
    dropout / dropout2 of nn.TransformerEncoderLayer): ``pai_dropout2d`` with one pixel per sample."""

    @staticmethod
    def forward(ctx, x, mask):
        _check_tokens(x)
        M, D = x.shape
        out = torch.empty_like(x)
        ops.dropout2d(x.dtype, x, mask, M, 1, D, out)
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        M, D = g.shape
        out = torch.empty_like(g)
        ops.dropout2d(g.dtype, g.contiguous(), mask, M, 1, D, out)
        return out, None
