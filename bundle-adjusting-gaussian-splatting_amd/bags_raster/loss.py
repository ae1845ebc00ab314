"""Photometric loss that produces dL/dimage for the op: 0.8*L1 + 0.2*(1-SSIM)  (train.py:311-313,325;
utils/loss_utils.py:18-76).  Plain PyTorch (host-side plumbing); the 11x11 Gaussian window is applied as two 1-D
passes, which is the same linear filter as the reference's dense 11x11 conv.  Pinned by tests/golden/loss.npz."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def l1_loss(x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return torch.abs(x - gt).mean()


def _window(size: int, sigma: float, like: torch.Tensor) -> torch.Tensor:
    g = torch.tensor([math.exp(-(i - size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(size)])
    return (g / g.sum()).to(like)


def _blur(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    c, n = x.shape[-3], w.numel()
    x = x if x.dim() == 4 else x.unsqueeze(0)
    x = F.conv2d(x, w.view(1, 1, n, 1).expand(c, 1, n, 1), padding=(n // 2, 0), groups=c)
    return F.conv2d(x, w.view(1, 1, 1, n).expand(c, 1, 1, n), padding=(0, n // 2), groups=c)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    w = _window(window_size, 1.5, img1)
    mu1, mu2 = _blur(img1, w), _blur(img2, w)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(img1 * img1, w) - mu1_sq
    s2 = _blur(img2 * img2, w) - mu2_sq
    s12 = _blur(img1 * img2, w) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def photometric_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))
