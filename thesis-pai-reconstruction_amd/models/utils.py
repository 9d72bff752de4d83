"""Metric / init helpers with the reference's names (reference models/utils.py:11-47).

``ssim`` / ``psnr`` / ``rmse`` / ``denormalize`` run on the HIP kernels
(csrc/ssim.hip, csrc/loss.hip); they raise on CPU tensors.
"""
import torch
import torch.nn as nn

from .. import functional as PF

denormalize = PF.denormalize          # reference models/utils.py:11
ssim = PF.ssim                        # reference models/utils.py:38-39
psnr = PF.psnr                        # reference models/utils.py:42-43
rmse = PF.rmse                        # reference models/utils.py:46-47


def to_int(x: torch.Tensor) -> torch.Tensor:
    """transforms.ConvertImageDtype(torch.uint8) on a float image in [0,1]
    (reference models/utils.py:12; torchvision 0.15.1: x * (255 + 1 - 1e-3), truncated)."""
    return (x * (255 + 1.0 - 1e-3)).to(torch.uint8)


def init_weights(module: nn.Module):
    """Reference models/utils.py:15-28: conv / linear weights ~ N(0, 0.02); norm affine (1, 0)."""
    if isinstance(module, (nn.Conv1d, nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
        nn.init.normal_(module.weight, 0.0, 0.02)
    if isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d, nn.GroupNorm, nn.LayerNorm)):
        nn.init.constant_(module.weight, 1.0)
        nn.init.constant_(module.bias, 0.0)


def get_parameter_count(model: nn.Module):
    """Reference models/utils.py:31-35."""
    if isinstance(model, nn.Module):
        return sum(p.numel() for p in model.parameters())
    return 0
