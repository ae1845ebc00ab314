"""bags_raster -- MI355X-native pose-differentiable Gaussian rasterizer (host side).

Mirrors the operator API of the reference's ``diff_gaussian_rasterization`` package
(gaussian_renderer/__init__.py:14,50-65,110-121): ``GaussianRasterizationSettings``, ``GaussianRasterizer``.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians, debug_views,
                         compute_relocation)

from .render import render, PipelineParams
from .gaussians import GaussianBag, eval_sh
from .io import save_ply, load_ply, save_checkpoint, load_checkpoint

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "debug_views",
           "compute_relocation", "render", "PipelineParams", "GaussianBag", "eval_sh",
           "save_ply", "load_ply", "save_checkpoint", "load_checkpoint"]
