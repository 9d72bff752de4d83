"""Attention U-Net generator (reference models/attention_unet.py:8-221) on the MI355X kernels.

Same module tree as the reference -- ``encoders`` / ``decoders`` / ``attention_blocks`` ModuleLists
whose leaves are stock ``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` parameter
containers -- so state-dict keys, shapes and initialisation are interchangeable.  ``forward`` hands
the network to ``AttentionUnetEngine``.
"""
from typing import Literal

import torch
import torch.nn as nn

from .. import functional as PF
from ..attention import AttentionUnetEngine
from .pix2pix import DecoderBlock, EncoderBlock
from .wrapper import UnetWrapper


class AttentionUnetGAN(UnetWrapper):
    """pix2pix with attention gates in the skip connections (Oktay et al. 2018); reference
    models/attention_unet.py:8-45."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 channel_mults=(1, 2, 4, 8, 8, 8, 8, 8), dropout: float = 0.5,
                 loss_type: Literal["gan", "ssim", "psnr", "ssim+psnr", "mse"] = "gan"):
        unet = AttentionUnet(in_channels, out_channels, channel_mults=channel_mults, dropout=dropout)
        super().__init__(unet, loss_type=loss_type)
        self.example_input_array = torch.Tensor(2, in_channels, 256, 256)
        self.save_hyperparameters()


class AttentionBlock(nn.Module):
    """x * Sigmoid(BN(conv1x1(ReLU(BN(conv1x1(signal)) + BN(conv1x1(x)))))) (reference
    models/attention_unet.py:48-96).  Parameter container; the arithmetic runs in AttentionUnetEngine."""

    def __init__(self, input_channels: int, signal_channels: int, attention_channels: int):
        super().__init__()
        self.input_gate = nn.Sequential(nn.Conv2d(input_channels, attention_channels, kernel_size=1),
                                        nn.BatchNorm2d(attention_channels))
        self.signal_gate = nn.Sequential(nn.Conv2d(signal_channels, attention_channels, kernel_size=1),
                                         nn.BatchNorm2d(attention_channels))
        self.attention = nn.Sequential(nn.Conv2d(attention_channels, 1, kernel_size=1), nn.BatchNorm2d(1),
                                       nn.Sigmoid())
        self.relu = nn.ReLU()


class AttentionUnet(nn.Module):
    """U-net with attention gates on the skip connections (reference models/attention_unet.py:99-221).

    :input: [N x in_channels x H x W]   :output: [N x out_channels x H x W]
    """

    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 channel_mults=(1, 2, 4, 8, 8, 8, 8, 8), dropout: float = 0.5):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.channel_mults = tuple(channel_mults)
        self.dropout = dropout
        self.compute_dtype = torch.float32
        self.bn_updates_per_forward = 1

        encoders = [nn.Conv2d(in_channels, channel_mults[0] * 64, kernel_size=4, stride=2, padding=1)]
        cin = channel_mults[0] * 64
        for level, mult in enumerate(channel_mults[1:], 1):
            channels = mult * 64
            encoders.append(EncoderBlock(cin, channels, norm=level != len(channel_mults) - 1))
            cin = channels
        self.encoders = nn.ModuleList(encoders)

        decoders, attention_blocks = [], []
        for level, mult in reversed(list(enumerate(channel_mults[:-1]))):
            channels = mult * 64
            decoders.append(DecoderBlock(
                cin, channels,
                dropout=dropout if (mult == max(channel_mults) and level > len(channel_mults) - 5) else 0,
            ))
            attention_blocks.append(AttentionBlock(channels, channels, channels // 2))
            cin = channels * 2
        decoders.append(nn.ConvTranspose2d(cin, out_channels, kernel_size=4, stride=2, padding=1))
        self.decoders = nn.ModuleList(decoders)
        self.attention_blocks = nn.ModuleList(attention_blocks)
        self.out = nn.Tanh()
        self._engine = None

    @property
    def supports_forward_reuse(self) -> bool:
        return not any(isinstance(m, nn.Dropout2d) and m.p > 0 for m in self.modules())

    @property
    def engine(self) -> AttentionUnetEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", AttentionUnetEngine(self))
        return self._engine

    def forward(self, x):
        eng = self.engine
        params = [p for p, _ in eng.ordered_params()]
        return PF.UnetFunction.apply(x, eng, self.training, self.bn_updates_per_forward,
                                     self.compute_dtype, *params)
