"""CPU restatement of the reference's photometric loss terms -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(bags_raster/loss.py -> csrc/loss.hip) never does.

PARITY PINNED: tests/test_golden_cpu.py checks these functions against tests/golden/loss.npz and loss_odd.npz, which
tests/golden/make_golden.py produced by running the reference's own utils/loss_utils.py (l1_loss, ssim and autograd) here.

Follows utils/loss_utils.py:
  :18-19   l1_loss  = mean |x - gt|
  :35-37   gaussian(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)), as float32, divided by its float32 sum
  :40-44   create_window: the 2-D window is the outer product of the 1-D window, one copy per channel
  :58-76   _ssim: five zero-padded depthwise 11x11 correlations (mu1, mu2, E[x^2], E[y^2], E[xy]); C1 = 0.01^2,
           C2 = 0.03^2; map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)); mean over everything
numpy, float64 accumulation (the reference computes in float32; the golden vectors bound the difference), explicit
121-tap window -- deliberately NOT the separable form the HIP kernels use.  The gradient is the analytic adjoint.
"""
import math

import numpy as np

C1 = 0.01 ** 2
C2 = 0.03 ** 2


def window_1d(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    g = np.array([math.exp(-(i - size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(size)], dtype=np.float32)
    return g / g.sum(dtype=np.float32)


def window_2d(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    g = window_1d(size, sigma)
    return (g[:, None] * g[None, :]).astype(np.float32)


def _correlate(x: np.ndarray, w2: np.ndarray) -> np.ndarray:
    """Zero-padded 'same' correlation of every (H,W) plane of x (C,H,W) with the 2-D window."""
    n = w2.shape[0]; r = n // 2
    C, H, W = x.shape
    xp = np.zeros((C, H + 2 * r, W + 2 * r), dtype=np.float64)
    xp[:, r:r + H, r:r + W] = x
    out = np.zeros((C, H, W), dtype=np.float64)
    for i in range(n):
        for j in range(n):
            out += float(w2[i, j]) * xp[:, i:i + H, j:j + W]
    return out


def l1_loss(x: np.ndarray, gt: np.ndarray) -> float:
    return float(np.abs(x.astype(np.float64) - gt.astype(np.float64)).mean())


def ssim_terms(x: np.ndarray, y: np.ndarray):
    w2 = window_2d()
    x = x.astype(np.float64); y = y.astype(np.float64)
    mu1, mu2 = _correlate(x, w2), _correlate(y, w2)
    e11, e22, e12 = _correlate(x * x, w2), _correlate(y * y, w2), _correlate(x * y, w2)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    return dict(mu1=mu1, mu2=mu2, A1=A1, A2=A2, B1=B1, B2=B2, map=(A1 * A2) / (B1 * B2), w2=w2)


def ssim(x: np.ndarray, y: np.ndarray) -> float:
    return float(ssim_terms(x, y)["map"].mean())


def loss_and_grad(x: np.ndarray, y: np.ndarray, g_l1: float, g_ssim: float):
    """(l1, ssim, d(g_l1 * l1 + g_ssim * ssim)/dx)."""
    t = ssim_terms(x, y)
    m, mu1, mu2, A1, A2, B1, B2, w2 = t["map"], t["mu1"], t["mu2"], t["A1"], t["A2"], t["B1"], t["B2"], t["w2"]
    n = x.size
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    # m as a function of (mu1, E11, E12) with s1 = E11 - mu1^2, s12 = E12 - mu1 mu2
    d_mu = 2 * mu2 * (A2 - A1) / (B1 * B2) + 2 * mu1 * m * (1 / B2 - 1 / B1)
    d_e11 = -m / B2
    d_e12 = 2 * A1 / (B1 * B2)
    w2t = w2[::-1, ::-1]               # adjoint of a correlation (the window is symmetric; kept for clarity)
    grad_ssim = _correlate(d_mu, w2t) + 2 * x64 * _correlate(d_e11, w2t) + y64 * _correlate(d_e12, w2t)
    grad = (g_ssim / n) * grad_ssim + (g_l1 / n) * np.sign(x64 - y64)
    return l1_loss(x, y), float(m.mean()), grad
