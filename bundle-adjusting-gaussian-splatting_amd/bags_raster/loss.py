"""Photometric loss that produces dL/dimage for the op: 0.8*L1 + 0.2*(1-SSIM)  (train.py:311-313,325;
utils/loss_utils.py:18-76).

Two implementations with the reference's names and semantics:
  * ``l1_loss / ssim / photometric_loss``        plain PyTorch (any device); the 11x11 Gaussian window is applied as two
    1-D passes, which is the same linear filter as the reference's dense 11x11 conv.
  * ``fused_l1_ssim / fused_photometric_loss``   the HIP kernels of csrc/loss.hip behind ``bags_loss_forward`` /
    ``bags_loss_backward`` (include/bags_raster.h): one forward and one backward kernel instead of ~50 launches.
    GPU tensors only; there is no fallback.
Both are pinned by tests/golden/loss.npz (values and dL/dimage from the reference's own functions)."""
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


# ------------------------------------------------------------------------------------------------ fused HIP path
class _FusedL1SSIM(torch.autograd.Function):
    """(image, gt) -> (mean |image - gt|, mean SSIM), differentiable w.r.t. ``image``."""

    @staticmethod
    def forward(ctx, image: torch.Tensor, gt: torch.Tensor):
        from . import _lib as L
        for name, t in (("image", image), ("gt", gt)):
            if not t.is_cuda:
                raise RuntimeError(f"fused_l1_ssim: {name} must be a GPU tensor (the fused loss has no CPU path; "
                                   f"use bags_raster.loss.l1_loss / ssim on the host)")
            if t.dtype != torch.float32:
                raise RuntimeError(f"fused_l1_ssim: {name} must be float32, got {t.dtype}")
        if image.shape != gt.shape or image.dim() != 3:
            raise RuntimeError(f"fused_l1_ssim: expected two (C,H,W) tensors of one shape, got {tuple(image.shape)} and {tuple(gt.shape)}")
        lib = L.load()
        image, gt = image.contiguous(), gt.contiguous()
        Cn, H, W = image.shape
        with torch.cuda.device(image.device):
            nbytes = lib.bags_loss_workspace_size(Cn, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
            terms = torch.empty(2, dtype=torch.float32, device=image.device)
            stream = torch.cuda.current_stream().cuda_stream
            L.check(lib.bags_loss_forward(image.data_ptr(), gt.data_ptr(), Cn, H, W, ws.data_ptr(), nbytes,
                                          terms.data_ptr(), stream), "bags_loss_forward")
        ctx.save_for_backward(image, gt, ws)
        return terms[0], terms[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        from . import _lib as L
        image, gt, ws = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        lib = L.load()
        Cn, H, W = image.shape
        z = torch.zeros((), dtype=torch.float32, device=image.device)
        gt_terms = torch.stack([z if g_l1 is None else g_l1.to(torch.float32), z if g_ssim is None else g_ssim.to(torch.float32)]).contiguous()
        grad = torch.empty_like(image)
        with torch.cuda.device(image.device):
            stream = torch.cuda.current_stream().cuda_stream
            L.check(lib.bags_loss_backward(image.data_ptr(), gt.data_ptr(), Cn, H, W, ws.data_ptr(), ws.numel(),
                                           gt_terms.data_ptr(), grad.data_ptr(), stream), "bags_loss_backward")
        return grad, None


def fused_l1_ssim(image: torch.Tensor, gt: torch.Tensor):
    """(l1_loss(image, gt), ssim(image, gt)) of utils/loss_utils.py in one HIP kernel; both scalars carry gradient."""
    return _FusedL1SSIM.apply(image, gt)


def _check_pair(fn: str, image: torch.Tensor, gt: torch.Tensor) -> None:
    for name, t in (("image", image), ("gt", gt)):
        if not t.is_cuda:
            raise RuntimeError(f"{fn}: {name} must be a GPU tensor (the fused loss has no CPU path; "
                               f"use bags_raster.loss.l1_loss / ssim on the host)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{fn}: {name} must be float32, got {t.dtype}")
    if image.shape != gt.shape or image.dim() != 3:
        raise RuntimeError(f"{fn}: expected two (C,H,W) tensors of one shape, got {tuple(image.shape)} and {tuple(gt.shape)}")


class _FusedPhotometric(torch.autograd.Function):
    """(image, gt, lambda) -> (loss, L1 mean, SSIM mean) with train.py:325's combination formed inside the reduce kernel and its
    backward inside the backward kernel (``bags_photometric_loss_*``): two launches forward, one backward, no one-element
    PyTorch kernels around them.  Only ``loss`` carries gradient; the two terms are returned for logging."""

    @staticmethod
    def forward(ctx, image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float):
        from . import _lib as L
        _check_pair("fused_photometric_loss", image, gt)
        lib = L.load()
        image, gt = image.contiguous(), gt.contiguous()
        Cn, H, W = image.shape
        with torch.cuda.device(image.device):
            nbytes = lib.bags_loss_workspace_size(Cn, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
            out = torch.empty(3, dtype=torch.float32, device=image.device)
            stream = torch.cuda.current_stream().cuda_stream
            L.check(lib.bags_photometric_loss_forward(image.data_ptr(), gt.data_ptr(), Cn, H, W, ws.data_ptr(), nbytes,
                                                      float(lambda_dssim), out.data_ptr(), stream), "bags_photometric_loss_forward")
        ctx.save_for_backward(image, gt, ws)
        ctx.lambda_dssim = float(lambda_dssim)
        loss, l1, s = out[0], out[1], out[2]
        ctx.mark_non_differentiable(l1, s)
        ctx.set_materialize_grads(False)          # no zero-fill launches for the two logging terms' absent gradients
        return loss, l1, s

    @staticmethod
    def backward(ctx, g_loss, _g_l1, _g_ssim):
        from . import _lib as L
        image, gt, ws = ctx.saved_tensors
        if not ctx.needs_input_grad[0] or g_loss is None:
            return None, None, None
        lib = L.load()
        Cn, H, W = image.shape
        g = g_loss.to(torch.float32).contiguous()
        grad = torch.empty_like(image)
        with torch.cuda.device(image.device):
            stream = torch.cuda.current_stream().cuda_stream
            L.check(lib.bags_photometric_loss_backward(image.data_ptr(), gt.data_ptr(), Cn, H, W, ws.data_ptr(), ws.numel(),
                                                       ctx.lambda_dssim, g.data_ptr(), grad.data_ptr(), stream),
                    "bags_photometric_loss_backward")
        return grad, None, None


def fused_photometric_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2, return_terms: bool = False):
    """train.py:325: (1 - lambda) L1 + lambda (1 - SSIM) as one fused op.  ``return_terms``: also the two means (detached)."""
    loss, l1, s = _FusedPhotometric.apply(image, gt, lambda_dssim)
    return (loss, l1, s) if return_terms else loss
