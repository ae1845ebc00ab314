"""``render()`` -- the Python caller of the rasterizer.

Mirror of the reference's ``gaussian_renderer.render`` (gaussian_renderer/__init__.py:30-133): same positional
arguments, same choice of covariance path (``pipe.compute_cov3D_python``) and colour path (``hybrid`` /
``pipe.convert_SHs_python`` / rasterizer-side SH), same returned dictionary.  Differences, all deliberate:

  * tensors are created on the device of ``pc.get_xyz`` instead of a hard-coded ``"cuda"``;
  * the camera chain (three calls in the reference, :57,58,61) is evaluated once per tensor, not once per use;
  * ``global_alignment`` may be ``None`` (the reference always passes a pair, ``train.py:250``);
  * ``shift_factors`` may be ``None`` (= zeros(3)); one extra keyword, ``depth_key`` (decision D6 of DESIGN.md);
  * the two screen-space gradient sinks are leaf tensors (the reference: ``zeros + 0`` with ``retain_grad()``, :37-44).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import torch

from .gaussians import eval_sh
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


class PipelineParams(SimpleNamespace):
    """The three switches of arguments/__init__.py:67-72."""

    def __init__(self, convert_SHs_python: bool = False, compute_cov3D_python: bool = False, debug: bool = False):
        super().__init__(convert_SHs_python=convert_SHs_python, compute_cov3D_python=compute_cov3D_python, debug=debug)


_ZERO3 = {}        # device -> zeros(3): the `shift_factors=None` stand-in (read-only: one fill launch per device, not per call)


def quaternion_multiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """Hamilton product, (w,x,y,z)   (gaussian_renderer/__init__.py:19-28)."""
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=-1)


def _python_colors(pc, xyz, feats, campos: torch.Tensor, mlp_color) -> torch.Tensor:
    """SH -> RGB in Python: eval_sh on unit view directions, +0.5, clamp at 0  (gaussian_renderer/__init__.py:90-95)."""
    shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dir_pp = xyz - campos.unsqueeze(0)
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
    return rgb + mlp_color


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, mlp_color, shift_factors, hybrid: bool = True,
           scaling_modifier: float = 1.0, override_color: Optional[torch.Tensor] = None, iteration: Optional[int] = None,
           global_alignment=None, depth_key: str = "z"):
    """Signature and defaults of gaussian_renderer/__init__.py:30: ``mlp_color`` and ``shift_factors`` are positional and
    required, ``hybrid`` defaults to True (Python-side SH colours + ``mlp_color``).  ``shift_factors=None`` stands for the
    zero vector the reference keeps (train.py:125-126: its optimizer is never stepped)."""
    # activations: one fused launch when the container offers it (GaussianBag on a GPU), else the reference's properties
    # rasterizer-side SH colours: the two feature parameters go to the op as they are stored (shs = features_dc, shs_rest =
    # features_rest), without get_features' torch.cat; every other colour path needs the (P,K,3) tensor
    raster_sh = override_color is None and not (hybrid or pipe.convert_SHs_python)
    rest = getattr(pc, "_features_rest", None)
    split = (raster_sh and torch.is_tensor(rest) and torch.is_tensor(getattr(pc, "_features_dc", None)) and rest.dim() == 3
             and rest.shape[1] >= 1 and rest.is_cuda)
    if hasattr(pc, "activated"):                               # GaussianBag: one fused launch each way
        xyz, features, opacity, scaling, rotation = pc.activated(features=not split)
    else:                                                      # any container with the reference's properties (GaussianModel)
        xyz, opacity, scaling, rotation = pc.get_xyz, pc.get_opacity, pc.get_scaling, pc.get_rotation
        features = None if split else pc.get_features
    # zero tensors whose .grad receives the screen-space gradients (:37-44)
    # (leaves: `.grad` is populated as with the reference's `zeros + 0` / retain_grad() pair, without the two adds and the two
    # 6 MB gradient copies that pair costs per call)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    screenspace_points_densify = torch.zeros_like(xyz, requires_grad=True)

    ga = global_alignment if global_alignment is not None else (None, None)
    if hasattr(viewpoint_camera, "get_matrices"):              # one HIP launch (bags_raster.camera.PoseCamera)
        viewmatrix, projmatrix, intrinsic, campos = viewpoint_camera.get_matrices(ga[0], ga[1])
    else:                                                      # any camera object with the reference's four getters
        viewmatrix = viewpoint_camera.get_world_view_transform(ga[0], ga[1])
        intrinsic = viewpoint_camera.get_intrinsic()
        projmatrix = (viewmatrix.unsqueeze(0).bmm(intrinsic.unsqueeze(0))).squeeze(0)
        campos = viewmatrix.inverse()[3, :3]

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),        # the STATIC fov, not the learnable one (:47-48)
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewmatrix,
        projmatrix=projmatrix,
        intrinsic=intrinsic,
        sh_degree=pc.active_sh_degree,
        campos=campos,
        prefiltered=False,
        debug=pipe.debug,
        debug_iter=iteration,
        depth_key=depth_key,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = scaling, rotation

    shs = shs_rest = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif hybrid or pipe.convert_SHs_python:
        colors_precomp = _python_colors(pc, xyz, features, campos, mlp_color)
    elif split:
        shs, shs_rest = pc._features_dc, pc._features_rest
    else:
        shs = features

    if shift_factors is None:
        shift_factors = _ZERO3.get(xyz.device)
        if shift_factors is None:
            shift_factors = _ZERO3[xyz.device] = torch.zeros(3, device=xyz.device)

    rendered_image, radii, depth, weights, mean2D = rasterizer(
        means3D=xyz, means2D=screenspace_points, means2D_densify=screenspace_points_densify,
        shift_factors=shift_factors, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, **({"shs_rest": shs_rest} if shs_rest is not None else {}))

    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "viewspace_points_densify": screenspace_points_densify,
            "visibility_filter": radii > 0,
            "radii": radii,
            "depth": depth,
            "weights": weights,
            "means2D": mean2D}
