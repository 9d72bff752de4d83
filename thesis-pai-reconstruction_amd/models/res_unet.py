"""Residual U-Net generator family (reference models/res_unet.py:10-334) on the MI355X kernels.

Same module tree as the reference -- ``in_conv``, ``encoders`` / ``decoders`` ModuleLists of blocks whose
``conv_block`` / ``conv_skip`` Sequentials hold stock ``nn.Conv2d`` / ``nn.BatchNorm2d`` parameter containers,
``out`` -- so state-dict keys and shapes are interchangeable.  The arithmetic runs through the op-level autograd
bridges of ``nnops`` (fused conv -> BatchNorm -> activation blocks, MaxPool2d, nearest Upsample, residual sum).
Built: ``res_type`` "18", "50", "next" and "v2" (pre-activation blocks: BatchNorm on a tensor that no convolution
epilogue produced, ``nnops.BNAct``).
"""
from typing import Literal

import torch
import torch.nn as nn

from .. import nnops
from ..ops import ACT_NONE, ACT_RELU, ACT_TANH, PaiError
from .wrapper import UnetWrapper

ResType = Literal["18", "50", "v2", "next"]


class ResUnetGAN(UnetWrapper):
    """Residual U-net behind the GAN wrapper (reference models/res_unet.py:10-49)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, res_type: ResType = "18",
                 channel_mults=(1, 2, 4, 8, 8, 8, 8, 8), dropout: float = 0.5,
                 loss_type: Literal["gan", "ssim", "psnr", "ssim+psnr", "mse"] = "gan"):
        unet = ResUnet(in_channels, out_channels, res_type, channel_mults=channel_mults, dropout=dropout)
        super().__init__(unet, loss_type=loss_type)
        self.example_input_array = torch.Tensor(2, in_channels, 256, 256)
        self.save_hyperparameters()


class _Block(nn.Module):
    """Shared executor of the residual blocks: ``conv_block`` is a Sequential of Conv2d / BatchNorm2d / ReLU
    entries, ``conv_skip`` a Conv2d + BatchNorm2d pair or Identity; ``post_relu`` = ReLU behind the sum."""
    post_relu = False

    def run(self, x, ctx):
        mods = list(self.conv_block)
        x, xs = nnops.fork(x)           # residual branch / skip branch
        h, i = x, 0
        ident = isinstance(self.conv_skip, nn.Identity)
        post = ACT_RELU if self.post_relu else ACT_NONE
        pre = None          # (holder, BatchNorm2d, act) of the previous convolution when ITS normalisation is left to the next one
        while i < len(mods):
            conv = mods[i]
            assert isinstance(conv, nn.Conv2d)
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
            j = i + (2 if bn is not None else 1)
            act = ACT_NONE
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                act, j = ACT_RELU, j + 1
            if j >= len(mods) and bn is not None and ctx.get("capture") is None and nnops.fuse_tail() \
                    and not (ident and isinstance(xs, tuple)):
                # last convolution of the block: its BatchNorm (+ ReLU), the skip branch's, the sum and the ReLU behind it in
                # one pass
                sk = (None, None) if ident else (self.conv_skip[0], self.conv_skip[1])
                return nnops.bn_tail(h, conv, bn, act, xs, sk[0], sk[1], post, ctx["training"], ctx["n_updates"], ctx["dtype"],
                                     pre=pre)
            if j < len(mods) and bn is not None and ctx.get("capture") is None \
                    and nnops.can_prologue(h, mods[j], ctx["dtype"], act):
                # the next convolution (and its weight gradient) reads this one's raw output through the BatchNorm + ReLU:
                # the activated tensor is never written (one tensor write and two reads less)
                hold = {}
                h = nnops.conv_bn_act(h, conv, bn, ACT_NONE, ctx["training"], ctx["n_updates"], ctx["dtype"], defer=hold, pre=pre)
                pre = (hold, bn, act)
                i = j
                continue
            h = nnops.conv_bn_act(h, conv, bn, act, ctx["training"], ctx["n_updates"], ctx["dtype"], pre=pre)
            pre = None
            if ctx.get("capture") is not None:
                ctx["capture"][f"{ctx['name']}.conv_block.{i}"] = h.detach().float()
            i = j
        if ident:
            s = nnops.as_tensor(xs)
        else:
            s = nnops.conv_bn_act(xs, self.conv_skip[0], self.conv_skip[1], ACT_NONE, ctx["training"], ctx["n_updates"],
                                  ctx["dtype"])
        return nnops.AddAct.apply(h, s, post)


def _skip(in_channels, out_channels):
    if in_channels == out_channels:
        return nn.Identity()
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1), nn.BatchNorm2d(out_channels))


class ResidualBlock18(_Block):
    """ResNet-18/34 block (reference models/res_unet.py:52-74): ReLU behind the sum."""
    post_relu = True

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels))
        self.conv_skip = _skip(in_channels, out_channels)
        self.out = nn.ReLU()


class ResidualBlock50(_Block):
    """ResNet-50 bottleneck block (reference models/res_unet.py:77-105)."""
    post_relu = True

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        bottleneck = in_channels // 4
        self.conv_block = nn.Sequential(
            nn.Conv2d(in_channels, bottleneck, kernel_size=1), nn.BatchNorm2d(bottleneck), nn.ReLU(),
            nn.Conv2d(bottleneck, bottleneck, kernel_size=3, padding=1), nn.BatchNorm2d(bottleneck), nn.ReLU(),
            nn.Conv2d(bottleneck, out_channels, kernel_size=1), nn.BatchNorm2d(out_channels))
        self.conv_skip = _skip(in_channels, out_channels)
        self.out = nn.ReLU()


class ResidualBlockV2(_Block):
    """Pre-activation block (reference models/res_unet.py:108-130): BatchNorm -> ReLU in FRONT of each convolution
    (standalone ``nnops.BNAct``: statistics of a stored tensor), bias-only convolutions, no ReLU behind the sum."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.BatchNorm2d(in_channels), nn.ReLU(), nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1))
        self.conv_skip = nn.Sequential(
            nn.BatchNorm2d(in_channels), nn.ReLU(), nn.Conv2d(in_channels, out_channels, kernel_size=1),
        ) if in_channels != out_channels else nn.Identity()

    def run(self, x, ctx):
        x = nnops.as_tensor(x)          # the BatchNorm in front needs the concatenation as one tensor
        tr, nu, dt = ctx["training"], ctx["n_updates"], ctx["dtype"]
        cb = self.conv_block
        h = nnops.BNAct.apply(x, cb[0].weight, cb[0].bias, cb[0], tr, nu, ACT_RELU)
        h = nnops.conv_bn_act(h, cb[2], None, ACT_NONE, tr, 0, dt)
        h = nnops.BNAct.apply(h, cb[3].weight, cb[3].bias, cb[3], tr, nu, ACT_RELU)
        h = nnops.conv_bn_act(h, cb[5], None, ACT_NONE, tr, 0, dt)
        if isinstance(self.conv_skip, nn.Identity):
            s = x
        else:
            sk = self.conv_skip
            s = nnops.BNAct.apply(x, sk[0].weight, sk[0].bias, sk[0], tr, nu, ACT_RELU)
            s = nnops.conv_bn_act(s, sk[2], None, ACT_NONE, tr, 0, dt)
        return nnops.AddAct.apply(h, s, ACT_NONE)


class ResidualBlockNeXt(_Block):
    """ResNeXt block (reference models/res_unet.py:133-171): 1x1 -> grouped 3x3 (groups = cardinality) -> 1x1, each
    with BatchNorm + ReLU (the last ReLU sits INSIDE conv_block), no ReLU behind the sum."""

    def __init__(self, in_channels: int, out_channels: int, cardinality: int = 32, bottleneck: int = 4):
        super().__init__()
        inner_width = bottleneck * cardinality
        self.conv_block = nn.Sequential(
            nn.Conv2d(in_channels, inner_width, kernel_size=1), nn.BatchNorm2d(inner_width), nn.ReLU(),
            nn.Conv2d(inner_width, inner_width, kernel_size=3, padding=1, groups=cardinality),
            nn.BatchNorm2d(inner_width), nn.ReLU(),
            nn.Conv2d(inner_width, out_channels, kernel_size=1), nn.BatchNorm2d(out_channels), nn.ReLU())
        self.conv_skip = _skip(in_channels, out_channels)


res_blocks = {"18": ResidualBlock18, "50": ResidualBlock50, "v2": ResidualBlockV2, "next": ResidualBlockNeXt}


class EncoderBlock(nn.Module):
    """Residual block -> MaxPool2d(2) (reference models/res_unet.py:182-203)."""

    def __init__(self, in_channels: int, out_channels: int, res_type: ResType):
        super().__init__()
        self.encode = nn.Sequential(res_blocks[res_type](in_channels, out_channels), nn.MaxPool2d(2))


class DecoderBlock(nn.Module):
    """Residual block -> Dropout2d | Identity -> Upsample(2) (reference models/res_unet.py:206-235)."""

    def __init__(self, in_channels: int, out_channels: int, res_type: ResType, dropout: float = 0.0):
        super().__init__()
        self.decode = nn.Sequential(res_blocks[res_type](in_channels, out_channels),
                                    nn.Dropout2d(dropout) if dropout > 0 else nn.Identity(),
                                    nn.Upsample(scale_factor=2))


class ResUnet(nn.Module):
    """Residual U-net (reference models/res_unet.py:238-334).

    :input: [N x in_channels x H x W]   :output: [N x out_channels x H x W]
    """

    def __init__(self, in_channels: int = 3, out_channels: int = 3, res_type: ResType = "18",
                 channel_mults=(1, 2, 4, 8, 8, 8, 8, 8), dropout: float = 0.5):
        super().__init__()
        self.in_channels, self.out_channels, self.res_type = in_channels, out_channels, res_type
        self.channel_mults = tuple(channel_mults)
        self.compute_dtype = torch.float32
        self.bn_updates_per_forward = 1
        self.dropout_mask_fn = None      # tests: fn(j, N, C, p, device) -> fp32 [N, C] of {0, 1 / (1 - p)}
        self.debug_capture = None        # tests: dict name -> detached NHWC activation
        self.in_conv = nn.Conv2d(in_channels, 64, kernel_size=3, padding=1)
        cin = 64
        encoders = []
        for level, mult in enumerate(channel_mults):
            channels = mult * 64
            encoders.append(EncoderBlock(cin, channels, res_type))
            cin = channels
        self.encoders = nn.ModuleList(encoders)
        decoders = []
        for level, mult in reversed(list(enumerate(channel_mults[:-1]))):
            channels = mult * 64
            decoders.append(DecoderBlock(
                cin, channels, res_type,
                dropout=dropout if (mult == max(channel_mults) and level > len(channel_mults) - 5) else 0))
            cin = channels * 2
        decoders.append(DecoderBlock(cin, channel_mults[0] * 64, res_type))
        self.decoders = nn.ModuleList(decoders)
        self.out = nn.Sequential(nn.Conv2d(channel_mults[0] * 64, out_channels, kernel_size=3, padding=1), nn.Tanh())

    @property
    def supports_forward_reuse(self) -> bool:
        return not any(isinstance(m, nn.Dropout2d) and m.p > 0 for m in self.modules())

    def forward(self, x):
        if not x.is_cuda:
            raise PaiError("ResUnet (HIP) needs a HIP device tensor; there is no CPU path")
        if x.shape[2] % (1 << len(self.encoders)) or x.shape[3] % (1 << len(self.encoders)):
            raise PaiError(f"input {x.shape[2]}x{x.shape[3]} must be divisible by 2^{len(self.encoders)}")
        dtype = self.compute_dtype
        ctx = {"training": self.training, "n_updates": self.bn_updates_per_forward, "dtype": dtype,
               "capture": self.debug_capture, "name": ""}
        nnops.prepack(self, dtype)          # the pointwise layers' bf16 packs: one launch for the whole pass
        h = nnops.to_nhwc(x, dtype)
        h = nnops.conv_bn_act(h, self.in_conv, None, ACT_NONE, self.training, 0, dtype)
        skips = []
        if self.debug_capture is not None:
            self.debug_capture["in"] = h.detach().float()
        for i, enc in enumerate(self.encoders):
            ctx["name"] = f"enc{i}"
            h = enc.encode[0].run(h, ctx)
            h = nnops.MaxPool2.apply(h)
            if i + 1 < len(self.encoders):
                h, sk = nnops.fork(h)     # next level / decoder
                skips.append(sk)
            else:
                skips.append(h)
            if self.debug_capture is not None:
                self.debug_capture[f"enc{len(skips) - 1}"] = h.detach().float()
        skips.pop()
        for j, dec in enumerate(self.decoders):
            if j != 0:
                h = (h, skips.pop())        # read as cat([h, skip]) by the block's convolutions (two-pointer inputs)
            ctx["name"] = f"dec{j}"
            h = dec.decode[0].run(h, ctx)
            drop = dec.decode[1]
            if self.training and isinstance(drop, nn.Dropout2d) and drop.p > 0:
                n, c = h.shape[0], h.shape[3]
                if self.dropout_mask_fn is not None:
                    mask = self.dropout_mask_fn(j, n, c, drop.p, h.device).to(torch.float32).contiguous()
                else:
                    mask = torch.bernoulli(torch.full((n, c), 1.0 - drop.p, device=h.device)) / (1.0 - drop.p)
                h = nnops.Dropout2d.apply(h, mask)
            h = nnops.Upsample2.apply(h)
            if self.debug_capture is not None:
                self.debug_capture[f"dec{j}"] = h.detach().float()
        pred = nnops.conv_bn_act(h, self.out[0], None, ACT_TANH, self.training, 0, dtype, out_f32=True)   # [N,H,W,Co] fp32
        return pred.permute(0, 3, 1, 2) if self.out_channels != 1 else pred.reshape(x.shape[0], 1, x.shape[2], x.shape[3])
