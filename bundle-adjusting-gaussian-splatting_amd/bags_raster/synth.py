"""Synthetic scenes and cameras for bench/tests.

``synth_scene`` follows the reference's own initialisation statistics: positions uniform in [-1.3,1.3]^3
(scene/dataset_readers.py:551), isotropic scale = expected nearest-neighbour distance (stand-in for
distCUDA2, scene/gaussian_model.py:177-178) x ``sm`` with log-normal anisotropy, random unit quaternions
(w,x,y,z), opacity sigmoid(logit(0.1)+N(0,1)) (scene/gaussian_model.py:182), DC colour via RGB2SH
(utils/sh_utils.py:115-116).  Tensors are the *activated* op inputs (scene/gaussian_model.py:118-141).
Everything is drawn on the CPU generator so every rank / device sees the same scene for a seed.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .camera import PoseCamera

SH_C0 = 0.28209479177387814
FOVY_DEFAULT = 0.6911112


def synth_scene(P: int, seed: int = 0, sm: float = 0.5, sh_degree: int = 3, device="cpu") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(P, 3, generator=g) * 2.6 - 1.3
    s_iso = sm * 0.554 * (P / 2.6 ** 3) ** (-1.0 / 3.0)
    scales = torch.exp(math.log(s_iso) + 0.3 * torch.randn(P, 3, generator=g))
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.sigmoid(math.log(0.1 / 0.9) + torch.randn(P, 1, generator=g))
    M = (sh_degree + 1) ** 2
    f_dc = (torch.rand(P, 1, 3, generator=g) - 0.5) / SH_C0
    if M > 1:
        f_rest = 0.05 * torch.randn(P, M - 1, 3, generator=g)
        shs = torch.cat([f_dc, f_rest], 1)
    else:
        shs = f_dc
    out = dict(means3D=xyz, scales=scales, rotations=rot, opacities=opacity, shs=shs.contiguous())
    return {k: v.to(device) for k, v in out.items()}


def look_at_origin_camera(width: int, height: int, dist: float = 4.0, fovy: float = FOVY_DEFAULT,
                          device="cpu", R: Optional[torch.Tensor] = None, T=None) -> PoseCamera:
    """R = I, T = (0,0,dist): camera ``dist`` units from the cloud centre looking down +z."""
    fovx = 2.0 * math.atan(width / height * math.tan(fovy / 2.0))
    R = torch.eye(3) if R is None else R
    T = torch.tensor([0.0, 0.0, dist]) if T is None else T
    return PoseCamera(R, T, fovx, fovy, width, height, device=device)


def so3_exp(w: torch.Tensor) -> torch.Tensor:
    """Rodrigues (utils/camera.py:58-73 uses the Taylor form; closed form here, values agree to fp32)."""
    th = w.norm()
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=w.dtype)
    if th < 1e-8:
        return torch.eye(3, dtype=w.dtype) + K
    return torch.eye(3, dtype=w.dtype) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * (K @ K)


def sphere_views(n: int, width: int, height: int, radius: float = 4.0, noise: float = 0.0, seed: int = 55,
                 device="cpu"):
    """n cameras on a radius-4 sphere (theta = 1.8 deg * k, phi = -30 deg; utils/pose_utils.py:59-64) looking at the
    origin, optionally perturbed like scene/__init__.py:121-148 (so3 / translation noise, generator seed 55)."""
    g = torch.Generator().manual_seed(seed)
    cams = []
    phi = math.radians(-30.0)
    for k in range(n):
        th = math.radians(1.8 * k)
        # camera centre on the sphere
        c = torch.tensor([radius * math.cos(phi) * math.sin(th), radius * math.sin(phi), -radius * math.cos(phi) * math.cos(th)])
        fwd = -c / c.norm()
        up = torch.tensor([0.0, -1.0, 0.0])
        right = torch.linalg.cross(up, fwd)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        R_c2w = torch.stack([right, down, fwd], 1)           # columns = camera axes in world
        if noise > 0:
            R_c2w = so3_exp(torch.randn(3, generator=g) * noise) @ R_c2w
        T = -(R_c2w.t() @ c)
        if noise > 0:
            T = T + torch.randn(3, generator=g) * noise
        cams.append(look_at_origin_camera(width, height, device=device, R=R_c2w, T=T))
    return cams
