"""torch.autograd bridges: PyTorch supplies the tape, libpai_hip.so supplies every number.

Each Function's forward/backward is a fixed sequence of C-ABI launches on the current
stream; nothing here synchronises with the host.
"""
from __future__ import annotations

import math

import torch

from . import ops


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
class _SlotRef:
    """Ownership of one activation slot of an engine between a forward and its backward.  The slot goes back to the
    engine's pool exactly once: after the (single) backward pass, or when the autograd context is dropped without
    one (a grad-enabled forward that is never backpropagated must not leak a full set of activations)."""

    def __init__(self, engine, slot):
        self.engine, self.slot = engine, slot

    def take(self):
        """The slot for the backward pass; a second backward through the same graph (retain_graph) is refused --
        the activations of the slot are overwritten in place by the first one."""
        if self.slot is None:
            raise ops.PaiError("second backward through the same forward: the engine's activation slot has already "
                               "been consumed (retain_graph is not supported)")
        slot, self.slot = self.slot, None
        return slot

    def __del__(self):
        if self.slot is not None:
            try:
                self.engine.release(self.slot)
            except Exception:      # interpreter shutdown
                pass
            self.slot = None


class UnetFunction(torch.autograd.Function):
    """Generator forward/backward through UnetEngine.  Parameter gradients are written into
    the engine's gradient arena and attached as ``p.grad`` directly (autograd receives None
    for them), so that data-parallel buckets can be reduced in place while the backward runs."""

    @staticmethod
    def forward(ctx, x, engine, training, bn_updates, dtype, *params):
        pred, slot = engine.forward(x, training, bn_updates, dtype)
        keep = any(ctx.needs_input_grad)
        if ctx.needs_input_grad[0]:
            raise ops.PaiError("gradient w.r.t. the generator input is not supported")
        if keep:
            ctx.engine, ctx.ref, ctx.params = engine, _SlotRef(engine, slot), params
        else:
            engine.release(slot)
        return pred

    @staticmethod
    def backward(ctx, gpred):
        engine, slot, params = ctx.engine, ctx.ref.take(), ctx.params
        arena = engine.arena()
        if getattr(engine, "overwrites_weight_grads", False):
            fresh = arena.begin_backward(params, overwrite_weights=True)
            engine.backward(slot, gpred, fresh)
        else:       # engines that ADD every gradient (attention U-Net): the whole arena is cleared
            arena.begin_backward(params)
            engine.backward(slot, gpred)
        arena.attach(params)
        engine.release(slot)
        return (None,) * (5 + len(params))


class DiscFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, engine, dtype, *params):
        logits, slot = engine.forward(x, y, dtype)
        if ctx.needs_input_grad[0]:
            raise ops.PaiError("gradient w.r.t. the conditioning image is not supported")
        if any(ctx.needs_input_grad):
            ctx.engine, ctx.ref, ctx.params = engine, _SlotRef(engine, slot), params
            ctx.need_dy = ctx.needs_input_grad[1]
            ctx.need_params = any(ctx.needs_input_grad[4:])
        else:
            engine.release(slot)
        return logits

    @staticmethod
    def backward(ctx, glogits):
        engine, slot, params = ctx.engine, ctx.ref.take(), ctx.params
        fresh = False
        if ctx.need_params:
            arena = engine.arena()
            fresh = arena.begin_backward(params, overwrite_weights=True)
        gy = engine.backward(slot, glogits, ctx.need_params, ctx.need_dy, fresh)
        if ctx.need_params:
            arena.attach(params)
        engine.release(slot)
        return (None, gy, None, None) + (None,) * len(params)


class DiscPairsFunction(torch.autograd.Function):
    """D(x, y_real) and D(x, y_fake) as ONE batch of 2N (the PatchGAN has no cross-sample coupling): returns the
    2N logits, real half first.  Only parameter gradients flow (the discriminator phase of the GAN step, reference
    models/wrapper.py:124-138: the generator output is detached there)."""

    @staticmethod
    def forward(ctx, x, y_real, y_fake, engine, dtype, *params):
        logits, slot = engine.forward(x, y_real, dtype, y2=y_fake)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise ops.PaiError("DiscPairsFunction carries no gradient to its image inputs; detach them")
        if any(ctx.needs_input_grad[5:]):
            ctx.engine, ctx.ref, ctx.params = engine, _SlotRef(engine, slot), params
        else:
            engine.release(slot)
        return logits

    @staticmethod
    def backward(ctx, glogits):
        engine, slot, params = ctx.engine, ctx.ref.take(), ctx.params
        arena = engine.arena()
        fresh = arena.begin_backward(params, overwrite_weights=True)
        engine.backward(slot, glogits, True, False, fresh)
        arena.attach(params)
        engine.release(slot)
        return (None,) * (5 + len(params))


class DiscGenLossFunction(torch.autograd.Function):
    """The generator's GAN loss in one node: ``BCE(D(x, pred), 1) + l1_weight * L1(pred, target)`` (reference
    models/wrapper.py:44-50).  ``pred`` feeds both terms; as two autograd nodes their gradients meet in an aten ``add``
    that torch launches itself -- the one kernel of the GAN step outside the C ABI, which a launch plan (plan.py) cannot
    contain.  Here the backward pass runs the discriminator's input gradient and adds the L1 term with ``pai_add_act``.
    Gradients flow to ``pred`` and, when they require them, to the discriminator's parameters."""

    @staticmethod
    def forward(ctx, x, pred, target, engine, dtype, l1_weight, *params):
        _check_f32_cuda(x, pred, target)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            raise ops.PaiError("gradient w.r.t. the conditioning image / the target is not supported")
        logits, slot = engine.forward(x, pred, dtype)
        pc, tc = pred.contiguous().float(), target.contiguous().float()
        need_pred, need_params = ctx.needs_input_grad[1], any(ctx.needs_input_grad[6:])
        need = need_pred or need_params
        gl = torch.empty_like(logits) if need else None
        gp = torch.empty_like(pc) if need_pred else None
        ent = _ACC.get(pc.device, "gan_g")
        ops.bce_logits(logits, 1.0, 1.0, ent[0], 1.0, gl)
        ops.l1(pc, tc, float(l1_weight), ent[0], float(l1_weight), gp)
        out = torch.empty((), dtype=torch.float32, device=pc.device)
        ops.scalar_take(ent[0], out)
        _ACC.taken(ent)
        if need:
            ctx.engine, ctx.ref, ctx.params = engine, _SlotRef(engine, slot), params
            ctx.need_pred, ctx.need_params = need_pred, need_params
            ctx.save_for_backward(*[g for g in (gl, gp) if g is not None])
        else:
            engine.release(slot)
        return out

    @staticmethod
    def backward(ctx, gout):
        engine, slot, params = ctx.engine, ctx.ref.take(), ctx.params
        saved = list(ctx.saved_tensors)
        gl = _scaled(saved.pop(0), gout)
        fresh = False
        if ctx.need_params:
            arena = engine.arena()
            fresh = arena.begin_backward(params, overwrite_weights=True)
        gy = engine.backward(slot, gl, ctx.need_params, ctx.need_pred, fresh)
        if ctx.need_params:
            arena.attach(params)
        engine.release(slot)
        if ctx.need_pred:
            gp = _scaled(saved.pop(0), gout)
            if gy.is_contiguous() and gy.numel() % 8 == 0:
                ops.add_act(torch.float32, gy, gp, ops.ACT_NONE, gy)
            else:
                gy = gy + gp
        return (None, gy, None, None, None, None) + (None,) * len(params)


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def _check_f32_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise ops.PaiError("pai losses/metrics need HIP device tensors (no CPU fallback exists)")


class _MeanLoss(torch.autograd.Function):
    """loss = mean-reduced {bce-with-logits vs a constant target, l1, mse}; the gradient is
    produced by the same kernel pass and scaled by grad_out in backward."""

    @staticmethod
    def forward(ctx, kind, x, target):
        _check_f32_cuda(x)
        xc = x.contiguous().float()
        acc = torch.zeros((), dtype=torch.float64, device=x.device)
        need = ctx.needs_input_grad[1]
        grad = torch.empty_like(xc) if need else None
        if kind == "bce":
            ops.bce_logits(xc, float(target), 1.0, acc, 1.0, grad)
        else:
            tc = target.contiguous().float()
            (ops.l1 if kind == "l1" else ops.mse)(xc, tc, 1.0, acc, 1.0, grad)
        if need:
            ctx.save_for_backward(grad)
        return acc.float()

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return None, _scaled(grad, gout), None


class _Accumulators:
    """Persistent fp64 device accumulators, zero between uses: the ``*_take`` launch that reads one also re-arms it,
    so a step spends no fill launch on it.  One per (device, stream, purpose); a use that did not reach its take
    (an exception in between) leaves the buffer dirty -- it is then replaced, not trusted."""

    def __init__(self):
        self.bufs = {}

    def get(self, device, purpose, n=1):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream, purpose)
        ent = self.bufs.get(key)
        if ent is None or ent[1] or ent[0].numel() != n:
            ent = [torch.zeros(n, dtype=torch.float64, device=device), False]
            self.bufs[key] = ent
        ent[1] = True          # armed: being accumulated into
        return ent

    @staticmethod
    def taken(ent):
        ent[1] = False


_ACC = _Accumulators()


_UNIT = {}


def unit_seed(device) -> torch.Tensor:
    """The cached fp32 scalar 1.0 of a device: ``LightningModule.manual_backward`` seeds scalar losses with it."""
    t = _UNIT.get(device.index)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _UNIT[device.index] = t
    return t


def _scaled(grad, gout):
    """grad * gout -- without the launch when gout IS the cached unit seed (a loss backpropagated directly)."""
    u = _UNIT.get(gout.device.index) if gout.is_cuda else None
    if u is not None and gout.data_ptr() == u.data_ptr():
        return grad
    return grad * gout


class _GanDiscLoss(torch.autograd.Function):
    """BCE(real logits, 1) + BCE(fake logits, 0), each mean-reduced (reference models/wrapper.py:68-95), over ONE
    tensor of 2N logits with the real half first: two loss launches accumulate into one fp64 scalar, a third turns it
    into the fp32 loss.  The gradient w.r.t. all 2N logits comes out of the same two passes."""

    @staticmethod
    def forward(ctx, labels, n_real):
        _check_f32_cuda(labels)
        x = labels.contiguous().float()
        flat = x.view(-1)
        k = (flat.numel() // x.shape[0]) * int(n_real)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(flat) if need else None
        ent = _ACC.get(x.device, "gan_d")
        ops.bce_logits(flat[:k], 1.0, 1.0, ent[0], 1.0, grad[:k] if need else None)
        ops.bce_logits(flat[k:], 0.0, 1.0, ent[0], 1.0, grad[k:] if need else None)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ops.scalar_take(ent[0], out)
        _ACC.taken(ent)
        if need:
            ctx.save_for_backward(grad)
            ctx.shape = labels.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, gout).view(ctx.shape), None


def gan_discriminator_loss_pairs(labels: torch.Tensor, n_real: int) -> torch.Tensor:
    """discriminator_loss(labels[n_real:], labels[:n_real]) of reference models/wrapper.py:68-95 for the logits of a
    batched (real | fake) discriminator pass."""
    return _GanDiscLoss.apply(labels, n_real)


class _GanGenLoss(torch.autograd.Function):
    """BCE(D(x, pred), 1) + l1_weight * L1(pred, target) (reference models/wrapper.py:44-50): both mean losses
    accumulate into one fp64 scalar (the L1 launch scaled by l1_weight), a third launch makes the fp32 value."""

    @staticmethod
    def forward(ctx, pred_label, pred, target, l1_weight):
        _check_f32_cuda(pred_label, pred, target)
        lc, pc, tc = pred_label.contiguous().float(), pred.contiguous().float(), target.contiguous().float()
        need_l, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gl = torch.empty_like(lc) if need_l else None
        gp = torch.empty_like(pc) if need_p else None
        ent = _ACC.get(pc.device, "gan_g")
        ops.bce_logits(lc, 1.0, 1.0, ent[0], 1.0, gl)
        ops.l1(pc, tc, float(l1_weight), ent[0], float(l1_weight), gp)
        out = torch.empty((), dtype=torch.float32, device=pc.device)
        ops.scalar_take(ent[0], out)
        _ACC.taken(ent)
        ctx.save_for_backward(*[g for g in (gl, gp) if g is not None])
        ctx.which = (need_l, need_p)
        return out

    @staticmethod
    def backward(ctx, gout):
        saved = list(ctx.saved_tensors)
        gl = _scaled(saved.pop(0), gout) if ctx.which[0] else None
        gp = _scaled(saved.pop(0), gout) if ctx.which[1] else None
        return gl, gp, None, None


def gan_generator_loss(pred_label, pred, target, l1_weight: float) -> torch.Tensor:
    """bce(D(x, pred), ones) + l1_weight * l1(pred, target): the ``loss_type == "gan"`` branch of reference
    models/wrapper.py:44-50."""
    return _GanGenLoss.apply(pred_label, pred, target, float(l1_weight))


def bce_with_logits_const(logits: torch.Tensor, target: float) -> torch.Tensor:
    """F.binary_cross_entropy_with_logits(logits, full_like(logits, target)) (reference
    models/wrapper.py:45-48,84-93)."""
    return _MeanLoss.apply("bce", logits, target)


def l1_loss(pred, target):
    """F.l1_loss (reference models/wrapper.py:49)."""
    return _MeanLoss.apply("l1", pred, target)


def mse_loss(pred, target):
    """F.mse_loss (reference models/wrapper.py:66)."""
    return _MeanLoss.apply("mse", pred, target)


class _Denormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _check_f32_cuda(x)
        xc = x.contiguous().float()
        out = torch.empty_like(xc)
        ops.denormalize(xc, None, out)
        ctx.save_for_backward(xc)
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        out = torch.empty_like(xc)
        ops.denormalize(xc, g.contiguous().float(), out)
        return out


def denormalize(x):
    """clamp(x*0.5+0.5, 0, 1) (reference models/utils.py:11)."""
    return _Denormalize.apply(x)


# --------------------------------------------------------------------------------------
# metrics (differentiable where the reference uses them as losses)
# --------------------------------------------------------------------------------------
def _ssim_sse(pred, target, denorm, per_image=False, full=False):
    _check_f32_cuda(pred, target)
    p = pred.contiguous().float()
    t = target.contiguous().float()
    n, c, h, w = p.shape
    out2 = torch.zeros(2, dtype=torch.float64, device=p.device)
    per = torch.zeros(n * c, dtype=torch.float64, device=p.device) if per_image else None
    fm = torch.empty(n, c, h, w, dtype=torch.float32, device=p.device) if full else None
    ops.ssim_sse(p, t, n * c, h, w, denorm, out2, per, fm)
    return p, t, out2, per, fm


class _SsimPsnr(torch.autograd.Function):
    """value = w_ssim * SSIM + w_psnr * PSNR (data_range 1) of (optionally denormalised) images."""

    @staticmethod
    def forward(ctx, pred, target, w_ssim, w_psnr, denorm):
        p, t, out2, _, _ = _ssim_sse(pred, target, denorm)
        n, c, h, w = p.shape
        ssim_v = out2[0] / (n * c)
        psnr_v = -torch.log(out2[1] / p.numel()) * (10.0 / math.log(10.0))
        ctx.save_for_backward(p, t, out2)
        ctx.cfg = (w_ssim, w_psnr, denorm)
        return (w_ssim * ssim_v + w_psnr * psnr_v).float()

    @staticmethod
    def backward(ctx, gout):
        p, t, out2 = ctx.saved_tensors
        w_ssim, w_psnr, denorm = ctx.cfg
        n, c, h, w = p.shape
        ws = torch.empty(ops.ssim_bwd_workspace_floats(n * c, h, w), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        # the kernel returns d[-(w_ssim*SSIM + w_psnr*PSNR)]/dpred
        ops.ssim_psnr_bwd(p, t, n * c, h, w, denorm, w_ssim, w_psnr, out2[1:], grad, ws)
        return grad * (-gout), None, None, None, None


def ssim(pred, target):
    """structural_similarity_index_measure(pred, target, data_range=1.0) (reference
    models/utils.py:38-39).  Inputs are already denormalised."""
    return _SsimPsnr.apply(pred, target, 1.0, 0.0, 0)


def psnr(pred, target):
    """peak_signal_noise_ratio(pred, target, data_range=1.0) (reference models/utils.py:42-43)."""
    return _SsimPsnr.apply(pred, target, 0.0, 1.0, 0)


def rmse(pred, target):
    """mean_squared_error(pred, target, squared=False) (reference models/utils.py:46-47)."""
    p, t, out2, _, _ = _ssim_sse(pred.detach(), target.detach(), 0)
    return torch.sqrt(out2[1] / p.numel()).float()


def ssim_psnr_of_normalized(pred, target, w_ssim, w_psnr):
    """w_ssim*SSIM + w_psnr*PSNR of denormalize(pred), denormalize(target) with the
    denormalisation fused into the metric kernel (differentiable w.r.t. pred)."""
    return _SsimPsnr.apply(pred, target, float(w_ssim), float(w_psnr), 1)


def metrics_of_normalized(pred, target):
    """(ssim, psnr, rmse) of the denormalised pair in ONE pass over the images, no graph
    (the per-step logging of reference models/wrapper.py:150-156,168-173)."""
    _check_f32_cuda(pred, target)
    p, t = pred.detach().contiguous().float(), target.detach().contiguous().float()
    n, c, h, w = p.shape
    ent = _ACC.get(p.device, "metrics", 2)
    ops.ssim_sse(p, t, n * c, h, w, 1, ent[0], None, None)
    out3 = torch.empty(3, dtype=torch.float32, device=p.device)
    ops.metrics_take(ent[0], n * c, p.numel(), out3)
    _ACC.taken(ent)
    return out3[0], out3[1], out3[2]


def ssim_per_image(pred, target, return_full_image=False):
    """reduction='none' SSIM (reference report.py:78-84,207-212): per-image values and,
    optionally, the un-cropped SSIM map."""
    p, t, out2, per, fm = _ssim_sse(pred.detach(), target.detach(), 0, per_image=True, full=return_full_image)
    n, c = p.shape[:2]
    vals = per.view(n, c).mean(dim=1).float()
    return (vals, fm) if return_full_image else vals
