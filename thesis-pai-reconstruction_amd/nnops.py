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

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_TANH


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
    groups > 1)."""
    cout, cig, kh, kw = weight.shape
    w = weight.detach().to(torch.float32).permute(0, 2, 3, 1)           # [Cout, kh, kw, Cin/g]
    if groups == 1:
        return w.contiguous()
    cog = cout // groups
    dense = torch.zeros(cout, kh, kw, cig * groups, dtype=torch.float32, device=weight.device)
    dv = dense.view(groups, cog, kh, kw, groups, cig)
    idx = torch.arange(groups, device=weight.device)
    dv[idx, :, :, :, idx, :] = w.reshape(groups, cog, kh, kw, cig)
    return dense


def _grad_from_fwd_pack(dw: torch.Tensor, weight: torch.Tensor, groups: int) -> torch.Tensor:
    cout, cig, kh, kw = weight.shape
    d = dw.view(cout, kh, kw, cig * groups)
    if groups > 1:
        cog = cout // groups
        idx = torch.arange(groups, device=dw.device)
        d = d.view(groups, cog, kh, kw, groups, cig)[idx, :, :, :, idx, :].reshape(cout, kh, kw, cig)
    return d.permute(0, 3, 1, 2).contiguous()


class ConvBNAct(torch.autograd.Function):
    """act(BatchNorm2d(Conv2d(x))) with k = 1 or 3 ("same"), optional groups, optional norm.

    Replaces the aten::convolution / native_batch_norm / relu (+ backward) calls behind the ``nn.Sequential``
    blocks of reference models/res_unet.py:58-64,86-95,147-163,66-69 and the bare convolutions at :265,308."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, training, n_updates, act, groups, dtype, out_f32):
        _check(x)
        N, H, W, Cin = x.shape
        Cout, _, k, _ = weight.shape
        d = ops.make_desc(dtype, 0, N, H, W, Cin, 0, Cout, 1, 0, 0, act if bn is None else ACT_NONE, kernel=k)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), x.device)
        wm = _dense_fwd_pack(weight, groups)
        if dtype == torch.float32:
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
        if bn is None:
            if out_f32:                       # final conv + tanh (reference :307-315): fp32 NCHW-compatible output
                out = torch.empty(N, H, W, Cout, dtype=torch.float32, device=x.device)
                ops.conv_fwd(d, x, None, wf, b32, y_f32=out)
            elif act == ACT_NONE:
                out = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
                ops.conv_fwd(d, x, None, wf, b32, y_raw=out)
            else:
                out = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
                ops.conv_fwd(d, x, None, wf, b32, y_act=out)
            ctx.save_for_backward(x, out, wd, weight)
            return out
        z = torch.empty(N, H, W, Cout, dtype=dtype, device=x.device)
        f32 = dict(dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        scale, shift = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        if training:
            rows = ops.conv_fwd_stats_rows(d)
            stats = torch.empty(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, **f32)
            ops.conv_fwd(d, x, None, wf, b32, y_raw=z, stats=stats)
            mom = bn.momentum if bn.momentum is not None else 0.1
            ops.bn_finalize(stats, rows, Cout, M, gamma.detach(), beta.detach(), float(bn.eps), float(mom), n_updates,
                            bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, rstd, scale, shift)
        else:
            ops.conv_fwd(d, x, None, wf, b32, y_raw=z)
            ops.bn_eval_coeffs(Cout, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, float(bn.eps),
                               scale, shift)
        out = torch.empty_like(z)
        ops.bn_apply(dtype, z, M, Cout, scale, shift, act, out)
        ctx.training = training
        ctx.save_for_backward(x, out, wd, weight, z, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, g):
        d, act, dtype = ctx.d, ctx.act, ctx.dtype
        N, H, W, Cin, Cout = d.N, d.H, d.W, d.C1, d.Cout
        M = N * H * W
        g = g.contiguous()
        dev = g.device
        f32 = dict(dtype=torch.float32, device=dev)
        dgamma = dbeta = None
        if not ctx.has_bn:
            x, out, wd, weight = ctx.saved_tensors
            dz = torch.empty(N, H, W, Cout, dtype=dtype, device=dev)
            if ctx.out_f32:
                ops.tanh_bwd(dtype, out, g.float(), None, dz) if act == ACT_TANH else ops.cast(g.float(), dz)
            elif act == ACT_NONE:
                dz = g
            else:
                ops.act_bwd(dtype, g, act, None, ACT_NONE, out, g.numel(), dz)
        else:
            x, out, wd, weight, z, mean, rstd, gamma = ctx.saved_tensors
            if not ctx.training:
                raise ops.PaiError("backward through an eval-mode BatchNorm block is not supported")
            du = torch.empty_like(z)
            part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * Cout, **f32)
            sums = torch.empty(2 * Cout, **f32)
            dgamma, dbeta = torch.zeros(Cout, **f32), torch.zeros(Cout, **f32)
            ops.bn_bwd_reduce(dtype, g, act, None, ACT_NONE, out if act != ACT_NONE else None, z, M, Cout, mean, rstd, du,
                              part, sums, dgamma, dbeta)
            dz = torch.empty_like(z)
            ops.bn_bwd_apply(dtype, du, z, M, Cout, mean, rstd, gamma.detach(), sums, dz)
        k = weight.shape[2]
        dw = torch.zeros(Cout * k * k * Cin, **f32)
        # a conv bias in front of a BatchNorm has an identically zero gradient
        dbias = torch.zeros(Cout, **f32) if ctx.needs_input_grad[2] else None
        ops.conv_wgrad(d, x, None, dz, dw, dbias if (dbias is not None and not ctx.has_bn) else None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(N, H, W, Cin, dtype=dtype, device=dev)
            ops.conv_dgrad(d, dz, wd, dx, None)
        gw = _grad_from_fwd_pack(dw, weight, ctx.groups)
        return dx, gw, dbias, dgamma, dbeta, None, None, None, None, None, None, None


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


def conv_bn_act(x, conv, bn, act, training, n_updates, dtype, out_f32=False):
    """Run an ``nn.Conv2d`` (+ ``nn.BatchNorm2d``) parameter container through ConvBNAct."""
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    return ConvBNAct.apply(x, conv.weight, conv.bias, gamma, beta, bn, training, n_updates, act, conv.groups, dtype,
                           out_f32)
