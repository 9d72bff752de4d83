"""Oracle: the two-phase GAN training step and the plain (non-GAN) step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
``UnetWrapper.training_step`` (models/wrapper.py:117-162) with the
pytorch_lightning 2.0.2 manual-optimisation semantics it relies on
(toggle_optimizer = freeze every parameter not owned by the active optimiser),
``UnetWrapper.loss`` (:42-66), ``discriminator_loss`` (:68-95) and
``configure_optimizers`` (:97-115, Adam lr 2e-4, betas (0.5, 0.999), eps 1e-7).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from .metrics_ref import denormalize, psnr, rmse, ssim
from .attention_ref import attention_unet_forward
from .pix2pix_ref import disc_forward
from .pix2pix_ref import unet_forward as _pix2pix_forward


def unet_forward(st, x, training=True, **kw):
    """Generator forward selected by the state's keys: Attention U-Net when it carries
    ``attention_blocks.*`` (models/attention_unet.py), the Pix2Pix U-Net otherwise."""
    if "attention_blocks.0.input_gate.0.weight" in st:
        return attention_unet_forward(st, x, training=training, **kw)
    if "vit_bottleneck.pos_embedding" in st:
        from .trans_unet_ref import trans_unet_forward
        return trans_unet_forward(st, x, training=training, **kw)
    if "in_conv.weight" in st:
        from .res_unet_ref import res_unet_forward
        return res_unet_forward(st, x, training=training, **kw)
    return _pix2pix_forward(st, x, training=training, **kw)

LR = 2e-4
BETAS = (0.5, 0.999)
ADAM_EPS = 1e-7
L1_WEIGHT = 50.0          # models/wrapper.py:51 (not the paper's 100)


def _is_param(k: str) -> bool:
    return not (k.endswith("running_mean") or k.endswith("running_var")
                or k.endswith("num_batches_tracked"))


def param_keys(st: OrderedDict):
    return [k for k in st if _is_param(k)]


@dataclass
class AdamState:
    step: int = 0
    exp_avg: dict = field(default_factory=dict)
    exp_avg_sq: dict = field(default_factory=dict)


def adam_step(st: OrderedDict, grads: dict, opt: AdamState,
              lr: float = LR, betas=BETAS, eps: float = ADAM_EPS):
    """torch.optim.Adam (no weight decay, no amsgrad) written out:
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    opt.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** opt.step
    bc2 = 1 - b2 ** opt.step
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            m = opt.exp_avg.setdefault(k, torch.zeros_like(st[k]))
            v = opt.exp_avg_sq.setdefault(k, torch.zeros_like(st[k]))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            st[k].addcdiv_(m, denom, value=-lr / bc1)


def discriminator_loss(pred_label, target_label):
    """models/wrapper.py:68-95."""
    return (F.binary_cross_entropy_with_logits(pred_label, torch.zeros_like(pred_label))
            + F.binary_cross_entropy_with_logits(target_label, torch.ones_like(pred_label)))


def generator_loss(loss_type, d_st, x, pred, target):
    """models/wrapper.py:42-66."""
    if loss_type == "gan":
        pred_label = disc_forward(d_st, x, pred)
        bce = F.binary_cross_entropy_with_logits(pred_label, torch.ones_like(pred_label))
        return bce + L1_WEIGHT * F.l1_loss(pred, target)
    if loss_type == "ssim":
        return -ssim(denormalize(pred), denormalize(target))
    if loss_type == "psnr":
        return -psnr(denormalize(pred), denormalize(target))
    if loss_type == "ssim+psnr":
        return -(30 * ssim(denormalize(pred), denormalize(target))
                 + psnr(denormalize(pred), denormalize(target)))
    if loss_type == "mse":
        return F.mse_loss(pred, target)
    raise ValueError(loss_type)


def _with_grad(st, keys):
    leaves = {}
    view = OrderedDict(st)
    for k in keys:
        leaves[k] = st[k].detach().clone().requires_grad_(True)
        view[k] = leaves[k]
    return view, leaves


def gan_training_step(g_st, d_st, opt_g: AdamState, opt_d: AdamState, x, target,
                      loss_type: str = "gan", return_grads: bool = False, dropout: float = 0.0,
                      mask_log=None, masks=None):
    """One ``training_step`` (models/wrapper.py:117-162).  Mutates g_st/d_st/opt_*
    in place; returns the logged scalars (and optionally the gradients).

    ``dropout``: the generator's Dropout2d rate (two forwards = two independent mask draws from the global
    CPU generator, as in the reference); ``mask_log`` collects (decoder, mask) in draw order, ``masks``
    replays a list of masks instead of drawing.

    D phase (:120-138): G forward with G frozen (no graph, but BN running stats
    ARE updated -- SURVEY Q5/Q6), D(x,target), D(x,pred), d_loss, Adam(D).
    G phase (:140-162): G forward again (second BN running-stat update), loss
    against the *updated* D (Q7), metrics on grad-carrying pred (Q10), Adam(G).
    """
    logs = {}
    grads_out = {}
    if loss_type == "gan":
        with torch.no_grad():
            pred = unet_forward(g_st, x, training=True, dropout=dropout, mask_log=mask_log, masks=masks)
        dview, dleaves = _with_grad(d_st, param_keys(d_st))
        target_label = disc_forward(dview, x, target)
        pred_label = disc_forward(dview, x, pred)
        d_loss = discriminator_loss(pred_label, target_label)
        logs["d_loss"] = d_loss.detach().clone()
        dg = torch.autograd.grad(d_loss, list(dleaves.values()))
        dgrads = dict(zip(dleaves.keys(), dg))
        adam_step(d_st, dgrads, opt_d)
        if return_grads:
            grads_out["d"] = dgrads

    gview, gleaves = _with_grad(g_st, param_keys(g_st))
    pred = unet_forward(gview, x, training=True, dropout=dropout, mask_log=mask_log, masks=masks)
    loss = generator_loss(loss_type, d_st, x, pred, target)
    with torch.no_grad():
        dp, dt = denormalize(pred), denormalize(target)
        logs["loss"] = loss.detach().clone()
        logs["train_ssim"] = ssim(dp, dt)
        logs["train_psnr"] = psnr(dp, dt)
        logs["train_rmse"] = rmse(dp, dt)
    gg = torch.autograd.grad(loss, list(gleaves.values()), allow_unused=True)
    ggrads = dict(zip(gleaves.keys(), gg))
    adam_step(g_st, ggrads, opt_g)
    # BN buffers were updated on the views' shared tensors already (same objects).
    if return_grads:
        grads_out["g"] = ggrads
        grads_out["pred"] = pred.detach()
        return logs, grads_out
    return logs


def plain_training_step(g_st, opt_g, x, target, loss_type):
    return gan_training_step(g_st, None, opt_g, None, x, target, loss_type=loss_type)


def validation_step(g_st, x, target):
    """models/wrapper.py:164-173 -- eval-mode forward + metrics."""
    with torch.no_grad():
        pred = unet_forward(g_st, x, training=False)
        dp, dt = denormalize(pred), denormalize(target)
        return {"val_ssim": ssim(dp, dt), "val_psnr": psnr(dp, dt), "val_rmse": rmse(dp, dt)}
