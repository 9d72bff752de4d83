"""TransUNet generator (reference models/trans_unet.py:9-255) on the MI355X kernels.

Same module tree as the reference -- ``in_conv``, ``encoders`` (bottleneck residual blocks with a stride-2 3x3
convolution), ``vit_bottleneck`` (patch embedding, ``pos_embedding``, a stock ``nn.TransformerEncoder`` as the
parameter container of the 12 post-norm layers), ``decoders``, ``out`` -- so state-dict keys and shapes are
interchangeable.  The arithmetic runs through ``nnops``: convolution -> BatchNorm -> activation blocks, the
even-pixel subsample behind / in front of the stride-1 kernels for the strided convolutions, ``nn.Linear`` layers
as one-tap gather-GEMMs over the token rows, LayerNorm with the residual sum and the position embedding fused,
erf GELU and the attention core.

Reference behaviour kept on purpose: the encoder layers are built without ``batch_first`` but fed
``[n, patches, dim]``, so attention mixes the IMAGES of a batch at equal patch position (sequence = n).
``image_size`` is fixed at 256 by ``TransUnetGAN`` (:22): the patch grid is sized for 256 x 256 inputs.

``dropout`` > 0 (the class default is 0.5, the CLI default 0): the four Dropout sites of every encoder layer (attention
weights, behind the attention block, inside and behind the feed-forward block) draw their masks with
``torch.bernoulli`` on the device (or take them from ``dropout_mask_fn`` in tests); the kernels apply them.
"""
import math
from typing import Literal

import torch
import torch.nn as nn

from .. import nnops
from ..ops import ACT_NONE, ACT_RELU, ACT_TANH, PaiError
from .wrapper import UnetWrapper


class TransUnetGAN(UnetWrapper):
    """TransUNet behind the GAN wrapper (reference models/trans_unet.py:9-32)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, channel_mults=(1, 2, 2, 4, 4), patch_size: int = 2,
                 dropout: float = 0.5, loss_type: Literal["gan", "ssim", "psnr", "ssim+psnr", "mse"] = "gan"):
        unet = TransUnet(in_channels, out_channels, image_size=256, channel_mults=channel_mults, patch_size=patch_size,
                         num_heads=8, dropout=dropout)
        super().__init__(unet, loss_type=loss_type)
        self.example_input_array = torch.Tensor(2, in_channels, 256, 256)
        self.save_hyperparameters()


class EncoderBlock(nn.Module):
    """ResNet-50 style bottleneck block that halves the resolution (reference models/trans_unet.py:182-236):
    1x1 -> 3x3 stride 2 -> 1x1 (BatchNorm each, ReLU behind the first two) + 1x1 stride-2 skip, ReLU behind the sum."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        bottleneck = in_channels // 4
        self.decode = nn.Sequential(
            nn.Conv2d(in_channels, bottleneck, kernel_size=1, bias=False), nn.BatchNorm2d(bottleneck), nn.ReLU(),
            nn.Conv2d(bottleneck, bottleneck, kernel_size=3, stride=2, padding=1, bias=False),
            nn.BatchNorm2d(bottleneck), nn.ReLU(),
            nn.Conv2d(bottleneck, out_channels, kernel_size=1, bias=False), nn.BatchNorm2d(out_channels))
        self.skip = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=2, bias=False),
                                  nn.BatchNorm2d(out_channels))
        self.out = nn.ReLU()

    def run(self, x, ctx):
        tr, nu, dt = ctx["training"], ctx["n_updates"], ctx["dtype"]
        d = self.decode
        x, xs = nnops.fork(x)           # residual branch / strided 1x1 skip
        h = nnops.conv_bn_act(x, d[0], d[1], ACT_RELU, tr, nu, dt)
        h = nnops.conv_bn_act(h, d[3], None, ACT_NONE, tr, 0, dt)          # 3x3 at stride 1 ...
        h = nnops.Subsample2.apply(h)                                      # ... its even pixels are the stride-2 result
        h = nnops.BNAct.apply(h, d[4].weight, d[4].bias, d[4], tr, nu, ACT_RELU)
        if ctx.get("capture") is None and nnops.fuse_tail():      # BatchNorm of both branches + sum + ReLU in one pass
            return nnops.bn_tail(h, d[6], d[7], ACT_NONE, nnops.Subsample2.apply(xs), self.skip[0], self.skip[1], ACT_RELU,
                                 tr, nu, dt)
        h = nnops.conv_bn_act(h, d[6], d[7], ACT_NONE, tr, nu, dt)
        s = nnops.conv_bn_act(nnops.Subsample2.apply(xs), self.skip[0], self.skip[1], ACT_NONE, tr, nu, dt)
        return nnops.AddAct.apply(h, s, ACT_RELU)


class DecoderBlock(nn.Module):
    """(3x3 conv -> BatchNorm -> ReLU) x 2 -> nearest Upsample(2) (reference models/trans_unet.py:239-255)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.decode = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(),
            nn.Upsample(scale_factor=2))

    def run(self, x, ctx):
        tr, nu, dt = ctx["training"], ctx["n_updates"], ctx["dtype"]
        d = self.decode
        h = nnops.conv_bn_act(x, d[0], d[1], ACT_RELU, tr, nu, dt)
        h = nnops.conv_bn_act(h, d[3], d[4], ACT_RELU, tr, nu, dt)
        return nnops.Upsample2.apply(h)


class VisionTransformer(nn.Module):
    """Patch embedding + 12 post-norm transformer encoder layers (reference models/trans_unet.py:120-180)."""

    def __init__(self, channels: int, input_size: int, patch_size: int = 16, num_heads: int = 8, dropout: float = 0.5,
                 transformer_layers: int = 12):
        super().__init__()
        patch_dim = channels * patch_size * patch_size
        num_patches = (input_size ** 2) // (patch_size ** 2)
        self.patch_size, self.patch_dim, self.num_patches, self.num_heads = patch_size, patch_dim, num_patches, num_heads
        self.grid = int(math.sqrt(num_patches))
        self.dropout = dropout
        self.dropout_mask_fn = None      # tests: fn(site, layer, shape, p, device) -> mask of {0, 1 / (1 - p)}
        # index 0 is the reference's (parameter-free) Rearrange layer: the state-dict keys start at 1
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, patch_dim),
                                                nn.LayerNorm(patch_dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches, patch_dim))
        layer = nn.TransformerEncoderLayer(patch_dim, num_heads, dropout=dropout, activation="gelu")
        self.transformer = nn.TransformerEncoder(layer, transformer_layers, enable_nested_tensor=False)

    def run(self, h, ctx):
        """h: NHWC [n, s, s, c] -> NHWC of the same shape."""
        n, hs, ws, c = h.shape
        p, g = self.patch_size, self.grid
        if hs != g * p or ws != g * p or c * p * p != self.patch_dim:
            raise PaiError(f"ViT bottleneck built for {g * p}x{g * p}x{self.patch_dim // (p * p)} features, got "
                           f"{hs}x{ws}x{c} (TransUnetGAN fixes image_size = 256)")
        P, D = self.num_patches, self.patch_dim
        drop = self.training and self.dropout > 0

        def mask(site, li, *shape):
            """fp32 mask of {0, 1 / (1 - p)} for one Dropout site (RNG plumbing; the kernels apply it)."""
            pr = self.dropout
            if self.dropout_mask_fn is not None:
                return self.dropout_mask_fn(site, li, shape, pr, h.device).to(torch.float32).reshape(shape).contiguous()
            return torch.bernoulli(torch.full(shape, 1.0 - pr, device=h.device)) / (1.0 - pr)

        # "n c (h p1) (w p2) -> n (h w) (p1 p2 c)" on the NHWC tensor: data movement only
        t = nnops.SwapMid.apply(h, n * g, p, g, p * c).view(n * P, D)
        ln1, lin, ln2 = self.to_patch_embedding[1], self.to_patch_embedding[2], self.to_patch_embedding[3]
        t = nnops.LayerNorm.apply(t, None, ln1.weight, ln1.bias, ln1.eps, None)
        t = nnops.Linear.apply(t, lin.weight, lin.bias)
        t = nnops.LayerNorm.apply(t, None, ln2.weight, ln2.bias, ln2.eps, self.pos_embedding)      # ... += pos_embedding
        for li, layer in enumerate(self.transformer.layers):
            sa = layer.self_attn
            t, t_res = nnops.fork(t)    # sublayer / its residual
            qkv = nnops.Linear.apply(t, sa.in_proj_weight, sa.in_proj_bias)
            # sequence axis = the image batch
            a = nnops.MHACore.apply(qkv, n, P, self.num_heads, mask("attn", li, P * self.num_heads, n, n) if drop else None)
            a = nnops.Linear.apply(a, sa.out_proj.weight, sa.out_proj.bias)
            if drop:
                a = nnops.TokenDropout.apply(a, mask("sa", li, n * P, D))
            t = nnops.LayerNorm.apply(t_res, a, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps, None)
            t, t_res = nnops.fork(t)
            f = nnops.Linear.apply(t, layer.linear1.weight, layer.linear1.bias)
            f = nnops.GELU.apply(f)
            if drop:
                f = nnops.TokenDropout.apply(f, mask("ff", li, n * P, f.shape[1]))
            f = nnops.Linear.apply(f, layer.linear2.weight, layer.linear2.bias)
            if drop:
                f = nnops.TokenDropout.apply(f, mask("out", li, n * P, D))
            t = nnops.LayerNorm.apply(t_res, f, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, None)
            if ctx.get("capture") is not None:
                ctx["capture"][f"vit{li}"] = t.detach().float()
        # "n (h w) (p1 p2 c) -> n c (h p1) (w p2)", as NHWC
        return nnops.SwapMid.apply(t, n * g, g, p, p * c).view(n, hs, ws, c)


class TransUnet(nn.Module):
    """Trans U-net (reference models/trans_unet.py:35-117).

    :input: [N x in_channels x image_size x image_size]   :output: [N x out_channels x image_size x image_size]
    """

    def __init__(self, in_channels: int = 3, out_channels: int = 3, image_size: int = 256, channel_mults=(1, 2, 4, 8),
                 patch_size: int = 16, num_heads: int = 8, dropout: float = 0.5):
        super().__init__()
        self.in_channels, self.out_channels, self.image_size = in_channels, out_channels, image_size
        self.channel_mults = tuple(channel_mults)
        self.compute_dtype = torch.float32
        self.bn_updates_per_forward = 1
        self.debug_capture = None        # tests: dict name -> detached activation
        self.in_conv = nn.Conv2d(in_channels, 64, kernel_size=3, padding=1)
        cin = 64
        encoders = []
        for mult in channel_mults:
            channels = mult * 64
            encoders.append(EncoderBlock(cin, channels))
            cin = channels
        self.encoders = nn.ModuleList(encoders)
        self.vit_bottleneck = VisionTransformer(channels=channel_mults[-1] * 64,
                                                input_size=image_size // (2 ** len(channel_mults)),
                                                patch_size=patch_size, num_heads=num_heads, dropout=dropout,
                                                transformer_layers=12)
        decoders = []
        for mult in reversed(list(channel_mults[:-1])):
            channels = mult * 64
            decoders.append(DecoderBlock(cin, channels))
            cin = channels * 2
        decoders.append(DecoderBlock(cin, 64))
        self.decoders = nn.ModuleList(decoders)
        self.out = nn.Sequential(nn.Conv2d(64, out_channels, kernel_size=3, padding=1), nn.Tanh())

    @property
    def supports_forward_reuse(self) -> bool:
        return not (self.vit_bottleneck.dropout > 0)

    def forward(self, x):
        if not x.is_cuda:
            raise PaiError("TransUnet (HIP) needs a HIP device tensor; there is no CPU path")
        if x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise PaiError(f"TransUnet is built for {self.image_size}x{self.image_size} inputs, got {x.shape[2]}x{x.shape[3]}")
        dtype = self.compute_dtype
        ctx = {"training": self.training, "n_updates": self.bn_updates_per_forward, "dtype": dtype,
               "capture": self.debug_capture}
        cap = self.debug_capture
        h = nnops.to_nhwc(x, dtype)
        h = nnops.conv_bn_act(h, self.in_conv, None, ACT_NONE, self.training, 0, dtype)
        skips = []
        for i, enc in enumerate(self.encoders):
            h = enc.run(h, ctx)
            if i + 1 < len(self.encoders):
                h, sk = nnops.fork(h)     # next level / decoder
                skips.append(sk)
            else:
                skips.append(h)
            if cap is not None:
                cap[f"enc{i}"] = h.detach().float()
        skips.pop()
        h = self.vit_bottleneck.run(h, ctx)
        if cap is not None:
            cap["vit"] = h.detach().float()
        for j, dec in enumerate(self.decoders):
            if j != 0:
                h = (h, skips.pop())        # read as cat([h, skip]) by the decoder's first convolution (two-pointer input)
            h = dec.run(h, ctx)
            if cap is not None:
                cap[f"dec{j}"] = h.detach().float()
        pred = nnops.conv_bn_act(h, self.out[0], None, ACT_TANH, self.training, 0, dtype, out_f32=True)   # [N,H,W,Co] fp32
        return pred.permute(0, 3, 1, 2) if self.out_channels != 1 else pred.reshape(x.shape[0], 1, x.shape[2], x.shape[3])
