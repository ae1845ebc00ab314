"""Gaussian parameter container and the activations that feed the rasterizer.

Host-side mirror of the pieces of the reference's ``GaussianModel`` that sit on the hot path:

  * activations  ``get_xyz / get_scaling / get_rotation / get_opacity / get_features / get_covariance``
    (scene/gaussian_model.py:118-141; set up in ``setup_functions``, scene/gaussian_model.py:26-43)
  * ``build_rotation / build_scaling_rotation / strip_lowerdiag / strip_symmetric`` (utils/general_utils.py:114-163)
  * ``eval_sh / RGB2SH / SH2RGB`` (utils/sh_utils.py:57-122) -- the Python colour path of ``render()``
  * the consumers of the op's screen-space gradients, ``add_densification_stats``
    (scene/gaussian_model.py:449-455)

Everything here is device-agnostic torch (the reference hard-codes ``device="cuda"``); values are pinned by
tests/golden/{sh_basis,gaussian_activations}.npz, generated from the reference's own functions.
Densification, pruning, optimiser plumbing and PLY I/O are out of scope (not on the path).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis values, ``(..., (deg+1)^2)``, in the reference's ordering and sign convention
    (utils/sh_utils.py:26-43 constants, :73-110 polynomials)."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, SH_C0)]
    if deg > 0:
        cols += [-_SH_C1 * y, _SH_C1 * z, -_SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [_SH_C2[0] * xy, _SH_C2[1] * yz, _SH_C2[2] * (2.0 * zz - xx - yy), _SH_C2[3] * xz, _SH_C2[4] * (xx - yy)]
        if deg > 2:
            cols += [_SH_C3[0] * y * (3.0 * xx - yy), _SH_C3[1] * xy * z, _SH_C3[2] * y * (4.0 * zz - xx - yy),
                     _SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), _SH_C3[4] * x * (4.0 * zz - xx - yy),
                     _SH_C3[5] * z * (xx - yy), _SH_C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """``sh (..., C, K)`` coefficients, ``dirs (..., 3)`` unit directions -> ``(..., C)``  (utils/sh_utils.py:57-112)."""
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise ValueError(f"need {n} SH coefficients, got {sh.shape[-1]}")
    return (sh[..., :n] * sh_basis(deg, dirs).unsqueeze(-2)).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    return sh * SH_C0 + 0.5


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.log(x / (1.0 - x))


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """(N,4) quaternions (w,x,y,z), normalised here -> (N,3,3)   (utils/general_utils.py:129-152)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def build_scaling_rotation(s: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """L = R(q) diag(s)   (utils/general_utils.py:154-163)."""
    return build_rotation(r) * s.unsqueeze(1)


def strip_lowerdiag(L: torch.Tensor) -> torch.Tensor:
    """(N,3,3) -> (N,6) = xx, xy, xz, yy, yz, zz   (utils/general_utils.py:114-123)."""
    return torch.stack([L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]], dim=1)


def strip_symmetric(sym: torch.Tensor) -> torch.Tensor:
    return strip_lowerdiag(sym)


def covariance_from_scaling_rotation(scaling: torch.Tensor, scaling_modifier: float, rotation: torch.Tensor) -> torch.Tensor:
    """The reference's ``covariance_activation`` (scene/gaussian_model.py:27-31): strip(L L^T), L = R diag(mod * s)."""
    L = build_scaling_rotation(scaling_modifier * scaling, rotation)
    return strip_symmetric(L @ L.transpose(1, 2))


class GaussianBag:
    """The Gaussian set as the reference stores it: raw (pre-activation) leaves plus activation properties.

    ``_features_dc (P,1,3)`` and ``_features_rest (P,K-1,3)`` are concatenated along dim 1 by ``get_features``
    (scene/gaussian_model.py:131-134), which is the ``(P,K,3)`` layout the rasterizer's ``shs`` argument takes.
    """

    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e

    # ---- construction
    @classmethod
    def from_activated(cls, scene: Dict[str, torch.Tensor], sh_degree: int, device="cpu", requires_grad: bool = True) -> "GaussianBag":
        """Build from activated values (``synth_scene`` output): inverts the activations, as ``create_from_pcd`` does for
        its initial values (scene/gaussian_model.py:164-195)."""
        pc = cls(sh_degree)
        dev = torch.device(device)

        def leaf(t):
            return t.detach().to(dev, torch.float32).contiguous().requires_grad_(requires_grad)
        shs = scene["shs"]
        pc._xyz = leaf(scene["means3D"])
        pc._features_dc = leaf(shs[:, :1, :])
        pc._features_rest = leaf(shs[:, 1:, :])
        pc._scaling = leaf(torch.log(scene["scales"]))
        pc._rotation = leaf(scene["rotations"])
        pc._opacity = leaf(inverse_sigmoid(scene["opacities"]))
        P = pc._xyz.shape[0]
        pc.max_radii2D = torch.zeros(P, device=dev)
        pc.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        pc.denom = torch.zeros(P, 1, device=dev)
        pc.active_sh_degree = int(round(math.sqrt(shs.shape[1]))) - 1
        return pc

    def leaves(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    # ---- activations (scene/gaussian_model.py:118-141)
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier: float = 1.0):
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def activated(self, features: bool = True):
        """(get_xyz, get_features, get_opacity, get_scaling, get_rotation) in one HIP launch each way on a GPU
        (bags_activations_forward / _backward, csrc/activations.hip); the same properties evaluated one by one on the host.
        ``features=False``: no concatenation (second entry None) -- for the rasterizer's ``shs`` / ``shs_rest`` pair, which
        takes ``_features_dc`` and ``_features_rest`` as they are (bags_raster.render)."""
        if self._xyz.is_cuda:
            if not features:
                _, op, sc, rot = _FusedActivations.apply(None, None, self._opacity, self._scaling, self._rotation)
                return self._xyz, None, op, sc, rot
            shs, op, sc, rot = fused_activations(self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)
            return self._xyz, shs, op, sc, rot
        return self.get_xyz, (self.get_features if features else None), self.get_opacity, self.get_scaling, self.get_rotation

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- consumers of the op's screen-space gradients (scene/gaussian_model.py:449-455)
    def add_densification_stats(self, viewspace_point_tensor, viewspace_point_tensor_densify, update_filter, abs_grad: bool):
        if abs_grad:
            assert viewspace_point_tensor_densify is not None
            g = viewspace_point_tensor_densify.grad
        else:
            g = viewspace_point_tensor.grad
        self.xyz_gradient_accum[update_filter] += torch.norm(g[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1


class _FusedActivations(torch.autograd.Function):
    """``dc`` and ``rest`` may both be None (packed features): no SH concatenation, the first output is None."""

    @staticmethod
    def forward(ctx, dc, rest, opacity, scaling, rotation):
        from . import _lib as L
        feats = dc is not None
        if (dc is None) != (rest is None):
            raise RuntimeError("fused_activations: features_dc and features_rest are given together or not at all")
        ts = [None if t is None else t.detach().to(torch.float32).contiguous() for t in (dc, rest, opacity, scaling, rotation)]
        if not all(t.is_cuda for t in ts if t is not None):
            raise RuntimeError("fused_activations: tensors must live on a GPU (use the GaussianBag properties on the host)")
        P = ts[3].shape[0]
        K = 1 + ts[1].shape[1] if feats else 1
        if (feats and (ts[0].shape != (P, 1, 3) or ts[1].shape != (P, K - 1, 3))) or ts[2].numel() != P or ts[3].shape != (P, 3) or ts[4].shape != (P, 4):
            raise RuntimeError("fused_activations: expected features_dc (P,1,3), features_rest (P,K-1,3), opacity (P,1), scaling (P,3), rotation (P,4)")
        dev = ts[3].device
        shs = torch.empty(P, K, 3, dtype=torch.float32, device=dev) if feats else None
        op = torch.empty(P, 1, dtype=torch.float32, device=dev)
        sc = torch.empty(P, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(P, 4, dtype=torch.float32, device=dev)
        raw = L.BagsRawGaussians(P, K, *[None if t is None else t.data_ptr() for t in ts])
        lib = L.load()
        with torch.cuda.device(dev):
            L.check(lib.bags_activations_forward(raw, None if shs is None else shs.data_ptr(), op.data_ptr(), sc.data_ptr(), rot.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "bags_activations_forward")
        ctx.feats = feats
        ctx.save_for_backward(*[t for t in ts if t is not None])
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as 96 MB of zeros
        return shs, op, sc, rot

    @staticmethod
    def backward(ctx, g_shs, g_op, g_sc, g_rot):
        from . import _lib as L
        ts = list(ctx.saved_tensors)
        if not ctx.feats:
            ts = [None, None] + ts
        P = ts[3].shape[0]
        K = 1 + ts[1].shape[1] if ctx.feats else 1
        need = ctx.needs_input_grad
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_shs, g_op, g_sc, g_rot = c(g_shs), c(g_op), c(g_sc), c(g_rot)
        out = [torch.empty_like(ts[0]) if (ctx.feats and need[0] and g_shs is not None) else None,
               torch.empty_like(ts[1]) if (ctx.feats and need[1] and g_shs is not None) else None,
               torch.empty_like(ts[2]) if (need[2] and g_op is not None) else None,
               torch.empty_like(ts[3]) if (need[3] and g_sc is not None) else None,
               torch.empty_like(ts[4]) if (need[4] and g_rot is not None) else None]
        p = lambda t: None if t is None else t.data_ptr()
        raw = L.BagsRawGaussians(P, K, *[p(t) for t in ts])
        lib = L.load()
        with torch.cuda.device(ts[3].device):
            L.check(lib.bags_activations_backward(raw, p(g_shs), p(g_op), p(g_sc), p(g_rot), *[p(o) for o in out],
                                                  torch.cuda.current_stream().cuda_stream), "bags_activations_backward")
        return tuple(out)


def fused_activations(features_dc, features_rest, opacity, scaling, rotation):
    """(get_features, get_opacity, get_scaling, get_rotation) of scene/gaussian_model.py:118-141 in one HIP launch."""
    return _FusedActivations.apply(features_dc, features_rest, opacity, scaling, rotation)
