"""Pix2Pix generator (reference models/pix2pix.py:7-216) on the MI355X kernels.

``Unet`` keeps the reference's module tree -- ``encoders`` / ``decoders`` ModuleLists of
``EncoderBlock`` / ``DecoderBlock`` whose ``encode`` / ``decode`` Sequentials hold stock
``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.BatchNorm2d`` instances -- so parameter names,
shapes, initialisation and checkpoints are interchangeable with the reference.  Those leaf
modules are parameter containers only: ``Unet.forward`` hands the whole encoder-decoder to
``UnetEngine``, which schedules the HIP kernels.
"""
from typing import Literal

import torch
import torch.nn as nn

from .. import functional as PF
from ..engine import UnetEngine
from .wrapper import UnetWrapper


class Pix2Pix(UnetWrapper):
    """Implementation of pix2pix (Isola et al. 2018); reference models/pix2pix.py:7-43."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 channel_mults=(1, 2, 4, 8, 8, 8, 8, 8), dropout: float = 0.5,
                 loss_type: Literal["gan", "ssim", "psnr", "ssim+psnr", "mse"] = "gan"):
        unet = Unet(in_channels, out_channels, channel_mults=channel_mults, dropout=dropout)
        super().__init__(unet, loss_type=loss_type)
        self.example_input_array = torch.Tensor(2, in_channels, 256, 256)
        self.save_hyperparameters()


class EncoderBlock(nn.Module):
    """LeakyReLU(0.2) -> Conv2d(k4,s2,p1) -> BatchNorm2d | Identity (reference
    models/pix2pix.py:46-74)."""

    def __init__(self, in_channels: int, out_channels: int, norm: bool = True):
        super().__init__()
        self.encode = nn.Sequential(
            nn.LeakyReLU(0.2),
            nn.Conv2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1),
            nn.BatchNorm2d(out_channels) if norm else nn.Identity(),
        )


class DecoderBlock(nn.Module):
    """ReLU -> ConvTranspose2d(k4,s2,p1) -> BatchNorm2d -> Dropout2d | Identity (reference
    models/pix2pix.py:77-111)."""

    def __init__(self, in_channels: int, out_channels: int, dropout: float = 0.5):
        super().__init__()
        self.decode = nn.Sequential(
            nn.ReLU(),
            nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.Dropout2d(dropout) if dropout > 0 else nn.Identity(),
        )


class Unet(nn.Module):
    """U-net generator of the pix2pix GAN (reference models/pix2pix.py:114-216).

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

        decoders = []
        for level, mult in reversed(list(enumerate(channel_mults[:-1]))):
            channels = mult * 64
            decoders.append(DecoderBlock(
                cin, channels,
                # only the three widest decoder blocks carry dropout (reference :176-179)
                dropout=dropout if (mult == max(channel_mults) and level > len(channel_mults) - 5) else 0,
            ))
            cin = channels * 2
        decoders.append(nn.ConvTranspose2d(cin, out_channels, kernel_size=4, stride=2, padding=1))
        self.decoders = nn.ModuleList(decoders)
        self.out = nn.Tanh()
        self._engine = None

    @property
    def supports_forward_reuse(self) -> bool:
        """True when two forwards on the same batch are bit-identical (no dropout)."""
        return not any(isinstance(m, nn.Dropout2d) and m.p > 0 for m in self.modules())

    @property
    def engine(self) -> UnetEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", UnetEngine(self))
        return self._engine

    def forward(self, x):
        eng = self.engine
        params = [p for p, _ in eng.ordered_params()]
        return PF.UnetFunction.apply(x, eng, self.training, self.bn_updates_per_forward,
                                     self.compute_dtype, *params)
