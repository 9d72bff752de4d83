"""``UnetWrapper`` / ``Discriminator`` -- the reference's plugin surface
(reference models/wrapper.py:9-238) on the MI355X kernels.

Same class names, constructor signatures, attribute names, hooks and state-dict keys as the
reference.  Differences, all on this side of the boundary:
  * ``Discriminator`` receives ``in_channels`` from the generator instead of the reference's
    hard-coded default of 3, which cannot run on 1-channel data (Q1); key names are unchanged.
  * with dropout == 0 the two generator forwards of a GAN step are bit-identical, so ONE
    forward is run and the BatchNorm running statistics are updated twice (Q5/Q6).  Set
    ``reuse_generator_forward = False`` to execute the literal two-forward schedule.
"""
from typing import Literal

import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from ..engine import DiscEngine
from ..lightning import LightningModule
from .utils import denormalize, init_weights, psnr, rmse, ssim  # noqa: F401

L1_WEIGHT = 50  # reference models/wrapper.py:51


class UnetWrapper(LightningModule):
    """U-net wrapper with different loss functions (reference models/wrapper.py:9-173).

    :param unet: any ``nn.Module`` mapping [N x C x H x W] -> [N x C x H x W].
    :param loss_type: one of "gan", "ssim", "psnr", "mse", "ssim+psnr".
    """

    def __init__(self, unet: nn.Module,
                 loss_type: Literal["gan", "ssim", "psnr", "ssim+psnr", "mse"] = "gan"):
        super().__init__()
        self.automatic_optimization = False
        self.unet = unet
        self.loss_type = loss_type
        self.reuse_generator_forward = True

        self.discriminator = None
        if loss_type == "gan":
            self.discriminator = Discriminator(in_channels=getattr(unet, "in_channels", 1))
            self.discriminator.apply(init_weights)
        self.unet.apply(init_weights)

    def forward(self, x):
        return self.unet(x)

    def loss(self, x, pred, target):
        """Reference models/wrapper.py:42-66."""
        if self.loss_type == "gan":
            if pred.is_cuda and getattr(self.discriminator, "supports_fused_generator_loss", False):
                # D(x, pred) -> BCE + L1_WEIGHT * L1 as ONE autograd node: the two gradients w.r.t. pred are summed by a
                # C-ABI launch instead of autograd's own aten add (the step then contains no kernel of torch's)
                return self.discriminator.generator_loss(x, pred, target, L1_WEIGHT)
            pred_label = self.discriminator(x, pred)
            if pred.is_cuda:
                # bce(pred_label, ones) + L1_WEIGHT * l1(pred, target) in three launches, no tensor-op glue
                return PF.gan_generator_loss(pred_label, pred, target, L1_WEIGHT)
            bce_loss = PF.bce_with_logits_const(pred_label, 1.0)
            l1_loss = PF.l1_loss(pred, target)
            return bce_loss + L1_WEIGHT * l1_loss
        if self.loss_type == "ssim":
            return -PF.ssim_psnr_of_normalized(pred, target, 1.0, 0.0)
        if self.loss_type == "psnr":
            return -PF.ssim_psnr_of_normalized(pred, target, 0.0, 1.0)
        if self.loss_type == "ssim+psnr":
            return -PF.ssim_psnr_of_normalized(pred, target, 30.0, 1.0)
        if self.loss_type == "mse":
            return PF.mse_loss(pred, target)
        raise ValueError(f"unknown loss_type {self.loss_type!r}")

    def discriminator_loss(self, pred_label: torch.Tensor, target_label: torch.Tensor) -> torch.Tensor:
        """Reference models/wrapper.py:68-95: fake -> 0, real -> 1."""
        pred_loss = PF.bce_with_logits_const(pred_label, 0.0)
        target_loss = PF.bce_with_logits_const(target_label, 1.0)
        return pred_loss + target_loss

    def configure_optimizers(self):
        """Reference models/wrapper.py:97-115 (Adam lr 2e-4, betas (0.5, 0.999), eps 1e-7).  Networks
        that run on the HIP engines get ``ArenaAdam`` -- a ``torch.optim.Adam`` subclass whose step is
        ONE fused pass over the flat parameter / gradient / moment arenas; anything else gets the
        stock optimizer."""
        from ..optim import make_adam
        opt_g = make_adam(self.unet, lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
        if self.discriminator is not None:
            opt_d = make_adam(self.discriminator, lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
            return opt_g, opt_d
        return opt_g

    def _can_reuse_forward(self):
        return (self.reuse_generator_forward and getattr(self.unet, "supports_forward_reuse", False)
                and self.unet.training)

    def _metrics_async(self, pred, target):
        if not pred.is_cuda:
            return None
        side = getattr(self, "_metrics_stream", None)
        if side is None:
            side = torch.cuda.Stream(device=pred.device)
            object.__setattr__(self, "_metrics_stream", side)
        ops.stream_wait_last(side, torch.cuda.current_stream())      # C-ABI edges: part of a recorded launch plan
        with torch.cuda.stream(side):
            return PF.metrics_of_normalized(pred, target)

    def _metrics_join(self, vals):
        cur = torch.cuda.current_stream()
        ops.stream_wait_last(cur, self._metrics_stream)
        for v in vals:
            v.record_stream(cur)   # allocated on the side stream's pool, consumed on this one
        return vals

    @staticmethod
    def _arm(opt):
        """The optimizer whose ``step()`` follows the coming backward pass may start updating while the pass is still
        running (``ArenaAdam.arm_streaming``); a no-op for every other optimizer."""
        arm = getattr(opt, "arm_streaming", None)
        if arm is not None:
            arm()

    def training_step(self, batch, batch_idx):
        """Reference models/wrapper.py:117-162 (manual optimisation, D step then G step)."""
        x, target = batch
        reuse = self.loss_type == "gan" and self._can_reuse_forward()
        pred_g = None
        metrics_early = None

        if self.loss_type == "gan":
            opt_d = self.optimizers()[1]
            if reuse:
                # one generator forward serves both phases; BN running stats advance twice (Q6)
                self.unet.bn_updates_per_forward = 2
                try:
                    pred_g = self.unet(x)
                finally:
                    self.unet.bn_updates_per_forward = 1

                # The per-step SSIM / PSNR / RMSE (reference models/wrapper.py:150-156) depend only on
                # (pred, target): they are issued now on a side stream, where their kernels fill the tail
                # of the discriminator backward instead of extending the generator phase; same values.
                metrics_early = self._metrics_async(pred_g, target)

            # Train discriminator.
            self.toggle_optimizer(opt_d)
            pred = pred_g.detach() if reuse else self.unet(x)
            if getattr(self.discriminator, "supports_batched_pairs", False):
                # the PatchGAN has no cross-sample coupling: D(x,target) and D(x,pred) are run as
                # one batch of 2N (one backward pass, so gradient buckets can be reduced while it runs)
                n = x.shape[0]
                labels = self.discriminator.forward_pairs(x, target, pred)
                # discriminator_loss(labels[n:], labels[:n]) without slicing the logits inside the autograd graph
                d_loss = PF.gan_discriminator_loss_pairs(labels, n)
            else:
                target_label = self.discriminator(x, target)
                pred_label = self.discriminator(x, pred)
                d_loss = self.discriminator_loss(pred_label, target_label)
            self.log("d_loss", d_loss, prog_bar=True)
            self.discriminator.zero_grad(set_to_none=True)
            self._arm(opt_d)
            self.manual_backward(d_loss)
            opt_d.step()
            self.untoggle_optimizer(opt_d)

        opt_g = self.optimizers()
        if isinstance(opt_g, list):
            opt_g = opt_g[0]

        # Train U-net
        self.toggle_optimizer(opt_g)
        pred = pred_g if reuse else self.unet(x)
        loss = self.loss(x, pred, target)

        self.log("loss", loss, prog_bar=True)
        if metrics_early is not None:
            s, p, r = self._metrics_join(metrics_early)
        else:
            s, p, r = PF.metrics_of_normalized(pred, target)      # (raises on host tensors: there is no CPU path)
        self.log("train_ssim", s, prog_bar=True)
        self.log("train_psnr", p, prog_bar=True)
        self.log("train_rmse", r, prog_bar=True)

        self.unet.zero_grad(set_to_none=True)
        self._arm(opt_g)
        self.manual_backward(loss)
        opt_g.step()
        self.untoggle_optimizer(opt_g)

    def validation_step(self, batch, batch_idx):
        """Reference models/wrapper.py:164-173."""
        x, target = batch
        pred = self.forward(x)
        s, p, r = PF.metrics_of_normalized(pred, target)
        self.log("val_ssim", s, prog_bar=True)
        self.log("val_psnr", p, prog_bar=True)
        self.log("val_rmse", r, prog_bar=True)


class DiscriminatorBlock(nn.Module):
    """Conv2d(k4,s2,p1) -> (InstanceNorm2d | Identity) -> LeakyReLU(0.2)
    (reference models/wrapper.py:176-209).  Inside ``Discriminator`` it is a parameter container whose arithmetic runs
    through DiscEngine; used on its own (``forward``) it runs through the op-level kernels -- including ``norm=True``
    (``nn.InstanceNorm2d``), which the reference's Discriminator never enables but its class offers."""

    def __init__(self, in_channels: int, out_channels: int, norm: bool = False):
        super().__init__()
        self.compute_dtype = torch.float32
        self.block = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1),
            nn.InstanceNorm2d(out_channels) if norm else nn.Identity(),
            nn.LeakyReLU(0.2),
        )

    def forward(self, x):
        """:input: [N x in_channels x H x W]   :output: [N x out_channels x H/2 x W/2] (fp32)"""
        from .. import nnops
        from ..ops import ACT_LRELU, ACT_NONE, PaiError
        if not x.is_cuda:
            raise PaiError("DiscriminatorBlock (HIP) needs a HIP device tensor; there is no CPU path")
        conv, norm = self.block[0], self.block[1]
        dtype = self.compute_dtype
        h = x.to(torch.float32).permute(0, 2, 3, 1).contiguous()
        h = nnops.ToStorage.apply(h, dtype)
        if isinstance(norm, nn.InstanceNorm2d):
            if norm.affine or norm.track_running_stats:
                raise PaiError("DiscriminatorBlock: InstanceNorm2d with affine parameters / running statistics is not built")
            h = nnops.ConvK4S2.apply(h, conv.weight, conv.bias, ACT_NONE, dtype)
            h = nnops.InstanceNormAct.apply(h, float(norm.eps), ACT_LRELU)
        else:
            h = nnops.ConvK4S2.apply(h, conv.weight, conv.bias, ACT_LRELU, dtype)
        return h.float().permute(0, 3, 1, 2)


class Discriminator(nn.Module):
    """PatchGAN discriminator (reference models/wrapper.py:212-238).

    :input x: [N x in_channels x H x W] conditioning image
    :input y: [N x in_channels x H x W] real or generated image
    :output: [N x 1 x (H/16 - 1) x (W/16 - 1)] logits
    """

    def __init__(self, in_channels: int = 3):
        super().__init__()
        self.in_channels = in_channels
        self.compute_dtype = torch.float32
        self.supports_batched_pairs = True
        self.supports_fused_generator_loss = True
        self.discriminator = nn.Sequential(
            DiscriminatorBlock(in_channels * 2, 64, norm=False),
            DiscriminatorBlock(64, 128),
            DiscriminatorBlock(128, 256),
            DiscriminatorBlock(256, 512),
            nn.Conv2d(512, 1, kernel_size=4, padding=1, bias=False),
        )
        self._engine = None

    @property
    def engine(self) -> DiscEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", DiscEngine(self))
        return self._engine

    def forward(self, x, y):
        eng = self.engine
        params = [p for p, _ in eng.ordered_params()]
        return PF.DiscFunction.apply(x, y, eng, self.compute_dtype, *params)

    def generator_loss(self, x, y, target, l1_weight):
        """``bce(forward(x, y), ones) + l1_weight * l1(y, target)`` (the "gan" branch of reference
        models/wrapper.py:44-50) as one autograd node (``PF.DiscGenLossFunction``)."""
        eng = self.engine
        params = [p for p, _ in eng.ordered_params()]
        return PF.DiscGenLossFunction.apply(x, y, target, eng, self.compute_dtype, float(l1_weight), *params)

    def forward_pairs(self, x, y_real, y_fake):
        """``forward(cat([x, x]), cat([y_real, y_fake]))`` without the concatenations; inputs carry no gradient."""
        eng = self.engine
        params = [p for p, _ in eng.ordered_params()]
        return PF.DiscPairsFunction.apply(x.detach(), y_real.detach(), y_fake.detach(), eng, self.compute_dtype, *params)
