"""Image-space distortion resampling of the rendered image (utils/util_distortion.py:271-311, the ``apply2gt == False``
branch of ``apply_distortion``; call site train.py:255-263).

``resample_image``        the HIP kernels of csrc/resample.hip (``bags_resample_forward`` / ``bags_resample_backward``):
                          control-flow upsample + grid_sample + centre crop + mask in one pass each way. GPU only.
``resample_image_torch``  the same pipeline with the PyTorch calls the reference makes (any device); used on the host.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F


def center_crop(t: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    """(N,C,H,W) -> (N,C,th,tw): the reference crops with a second grid_sample on an integer grid
    (utils/util_distortion.py:58-77)."""
    _, _, H, W = t.shape
    sy, sx = (H - th) // 2, (W - tw) // 2
    gy, gx = torch.meshgrid(torch.linspace(sy, sy + th - 1, th), torch.linspace(sx, sx + tw - 1, tw), indexing="ij")
    grid = torch.stack((gx, gy), 2).unsqueeze(0).to(t.device)
    grid = 2.0 * grid / torch.tensor([W - 1, H - 1], device=t.device, dtype=grid.dtype) - 1.0
    return F.grid_sample(t, grid.expand(t.shape[0], th, tw, 2).to(t.dtype), align_corners=True)


def resample_image_torch(image: torch.Tensor, ctrl_flow: torch.Tensor, flow_hw: Tuple[int, int], crop_hw: Tuple[int, int]):
    flow = ctrl_flow
    if tuple(flow.shape[:2]) != tuple(flow_hw):
        flow = F.interpolate(flow.permute(2, 0, 1).unsqueeze(0), size=tuple(flow_hw), mode="bilinear",
                             align_corners=False).permute(0, 2, 3, 1).squeeze(0)
    img = F.grid_sample(image.unsqueeze(0), flow.unsqueeze(0), mode="bilinear", padding_mode="zeros", align_corners=True)
    img = center_crop(img, crop_hw[0], crop_hw[1]).squeeze(0)
    mask = (~((img[0] == 0.0) & (img[1] == 0.0)).unsqueeze(0)).float()
    return img, mask


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, ctrl_flow, flow_hw, crop_hw):
        from . import _lib as L
        for name, t in (("image", image), ("ctrl_flow", ctrl_flow)):
            if not t.is_cuda:
                raise RuntimeError(f"resample_image: {name} must be a GPU tensor (use resample_image_torch on the host)")
        if image.dim() != 3 or ctrl_flow.dim() != 3 or ctrl_flow.shape[2] != 2:
            raise RuntimeError(f"resample_image: expected image (C,H,W) and flow (h,w,2), got {tuple(image.shape)}, {tuple(ctrl_flow.shape)}")
        img = image.detach().to(torch.float32).contiguous()
        ctl = ctrl_flow.detach().to(torch.float32).contiguous()
        Cn, H, W = img.shape
        h, w = ctl.shape[:2]
        Hf, Wf = int(flow_hw[0]), int(flow_hw[1])
        Hc, Wc = int(crop_hw[0]), int(crop_hw[1])
        out = torch.empty(Cn, Hc, Wc, dtype=torch.float32, device=img.device)
        mask = torch.empty(1, Hc, Wc, dtype=torch.float32, device=img.device)
        lib = L.load()
        with torch.cuda.device(img.device):
            L.check(lib.bags_resample_forward(img.data_ptr(), Cn, H, W, ctl.data_ptr(), h, w, Hf, Wf, Hc, Wc, out.data_ptr(),
                                              mask.data_ptr(), None, torch.cuda.current_stream().cuda_stream),
                    "bags_resample_forward")
        ctx.save_for_backward(img, ctl)
        ctx.dims = (Cn, H, W, h, w, Hf, Wf, Hc, Wc)
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, g_out, _g_mask):
        from . import _lib as L
        img, ctl = ctx.saved_tensors
        Cn, H, W, h, w, Hf, Wf, Hc, Wc = ctx.dims
        g_out = g_out.to(torch.float32).contiguous()
        g_img = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        g_ctl = torch.empty_like(ctl) if ctx.needs_input_grad[1] else None
        lib = L.load()
        with torch.cuda.device(img.device):
            nbytes = lib.bags_resample_workspace_size(H, W, Hc, Wc)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=img.device)
            L.check(lib.bags_resample_backward(img.data_ptr(), Cn, H, W, ctl.data_ptr(), h, w, Hf, Wf, Hc, Wc, g_out.data_ptr(),
                                               ws.data_ptr(), nbytes, None if g_img is None else g_img.data_ptr(),
                                               None if g_ctl is None else g_ctl.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "bags_resample_backward")
        return g_img, g_ctl, None, None


def resample_image(image: torch.Tensor, ctrl_flow: torch.Tensor, flow_hw: Tuple[int, int], crop_hw: Tuple[int, int]):
    """(warped image (C,Hc,Wc), mask (1,Hc,Wc)); gradients reach ``image`` (the rasterizer) and ``ctrl_flow`` (the lens net)."""
    return _Resample.apply(image, ctrl_flow, tuple(flow_hw), tuple(crop_hw))
