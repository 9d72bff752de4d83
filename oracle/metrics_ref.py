"""Oracle: SSIM / PSNR / RMSE / MSE as the reference computes them.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference delegates these to ``torchmetrics==0.11.4``
(requirements.txt:6; call sites models/utils.py:38-47, report.py:78-96,146,
207-212), whose source is NOT vendored under /root/reference and is not
installed in this environment.  The algorithm below restates the published
torchmetrics 0.11.4 ``structural_similarity_index_measure`` defaults
(gaussian_kernel=True, sigma=1.5, kernel_size=11, k1=0.01, k2=0.03,
reflect padding, 5-px crop before the per-image mean) and is pinned against
scikit-image 0.18.3 (tests/golden/ssim_skimage.npz, made by
oracle/gen_ssim_skimage.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SSIM_SIGMA = 1.5
SSIM_K1 = 0.01
SSIM_K2 = 0.03


def denormalize(x: torch.Tensor) -> torch.Tensor:
    """models/utils.py:11 -- clamp(x*0.5+0.5, 0, 1)."""
    return torch.clamp(x * 0.5 + 0.5, 0, 1)


def gaussian_1d(sigma: float = SSIM_SIGMA, dtype=torch.float32) -> torch.Tensor:
    ks = 2 * int(3.5 * sigma + 0.5) + 1                    # 11
    d = torch.arange((1 - ks) / 2, (1 + ks) / 2, 1, dtype=dtype)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def ssim_full(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0):
    """Returns (per_image [N], full_map [N,C,H,W]).

    per_image = mean over C and the 5-px-cropped map; full_map is the
    un-cropped map computed on the reflect-padded inputs
    (``return_full_image=True`` in report.py:78-84).
    """
    dtype = pred.dtype
    n, c, h, w = pred.shape
    g = gaussian_1d(dtype=dtype)
    ks = g.numel()
    pad = (ks - 1) // 2
    kern = torch.outer(g, g).expand(c, 1, ks, ks).contiguous()
    p = F.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t], dim=0)
    out = F.conv2d(stack, kern, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(n, dim=0)
    c1 = (SSIM_K1 * data_range) ** 2
    c2 = (SSIM_K2 * data_range) ** 2
    s_pp = e_pp - mu_p * mu_p
    s_tt = e_tt - mu_t * mu_t
    s_pt = e_pt - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / (
        (mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    cropped = full[..., pad:-pad, pad:-pad]
    per_image = cropped.reshape(n, -1).mean(-1)
    return per_image, full


def ssim(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """models/utils.py:38-39 -- mean over the batch of the per-image SSIM."""
    return ssim_full(pred, target)[0].mean()


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """models/utils.py:42-43 -- one population over the whole tensor:
    10/ln10 * (2 ln R - ln(sum((p-t)^2)/numel))."""
    sse = ((pred - target) ** 2).sum()
    return (2 * math.log(data_range) - torch.log(sse / pred.numel())) * (10 / math.log(10))


def mse(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ((pred - target) ** 2).sum() / pred.numel()


def rmse(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """models/utils.py:46-47 -- sqrt(sum((p-t)^2)/numel)."""
    return torch.sqrt(mse(pred, target))


def depth_ssim(preds: torch.Tensor, targets: torch.Tensor, num_depths: int = 16):
    """report.py:188-217 -- per-strip (mean, std) of the per-image SSIM."""
    out = []
    for xp, xt in zip(preds.chunk(num_depths, dim=2), targets.chunk(num_depths, dim=2)):
        s = ssim_full(xp, xt)[0]
        out.append((s.mean(), s.std()))
    return torch.tensor(out)
