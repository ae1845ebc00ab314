"""Shared builders for the tests: scene + camera -> the op's inputs/settings for both the oracle and the HIP op."""
import math

import torch

from bags_raster.synth import synth_scene, look_at_origin_camera
from oracle import raster_oracle as O


def camera_tensors(cam, device="cpu"):
    with torch.no_grad():
        return dict(viewmatrix=cam.get_world_view_transform().detach().to(device).contiguous(),
                    projmatrix=cam.get_full_proj_transform().detach().to(device).contiguous(),
                    intrinsic=cam.get_intrinsic().detach().to(device).contiguous(),
                    campos=cam.get_camera_center().detach().to(device).contiguous())


def oracle_settings(cam, sh_degree, bg=None, scale_modifier=1.0, depth_key="z", tile_bounds="opacity", clamp_grad="stock",
                    conic_grad="stock"):
    ct = camera_tensors(cam)
    return O.OracleSettings(image_height=cam.image_height, image_width=cam.image_width,
                            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                            bg=torch.zeros(3) if bg is None else bg, scale_modifier=scale_modifier,
                            sh_degree=sh_degree, depth_key=depth_key, tile_bounds=tile_bounds, clamp_grad=clamp_grad,
                            conic_grad=conic_grad, **ct)


def hip_settings(cam, sh_degree, device, bg=None, scale_modifier=1.0, depth_key="z", tensors=None, debug=False,
                 tile_bounds="opacity", binning="auto", clamp_grad="stock", conic_grad="stock"):
    from bags_raster import GaussianRasterizationSettings
    ct = tensors if tensors is not None else camera_tensors(cam, device)
    bg = torch.zeros(3) if bg is None else bg
    return GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width,
                                         tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                         bg=bg.to(device), scale_modifier=scale_modifier, viewmatrix=ct["viewmatrix"],
                                         projmatrix=ct["projmatrix"], intrinsic=ct["intrinsic"], sh_degree=sh_degree,
                                         campos=ct["campos"], prefiltered=False, debug=debug, debug_iter=0,
                                         depth_key=depth_key, tile_bounds=tile_bounds, binning=binning, clamp_grad=clamp_grad,
                                         conic_grad=conic_grad)


def make_case(P, W, H, sm=1.0, deg=3, seed=0, dist=4.0, **cam_kw):
    scene = synth_scene(P, seed, sm, deg)
    cam = look_at_origin_camera(W, H, dist=dist, **cam_kw)
    return scene, cam


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """|a-b|_2 / |b|_2 in fp64 (b = reference)."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    n = b.norm().item()
    return (a - b).norm().item() / n if n > 0 else (a - b).norm().item()
