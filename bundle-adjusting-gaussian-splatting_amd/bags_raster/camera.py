"""Pose -> matrix chain feeding GaussianRasterizationSettings (stays PyTorch).

Device-agnostic restatement of the reference's learnable camera:
  * ``projection_matrix``            <- utils/graphics_utils.py:83-107  (getProjectionMatrix)
  * ``quaternion_to_rotation``       <- scene/cameras.py:399-416
  * ``PoseCamera.world_view_transform / full_proj_transform / camera_center / intrinsic``
                                     <- scene/cameras.py:95-113, 356-381
The reference classes cannot be imported off-GPU (``.cuda()`` in constructors and default arguments), so the
bench/tests build their cameras here; values and Jacobians are pinned against the reference by tests/golden/
camera_chain.npz (getProjectionMatrix, quaternion_to_rotation_matrix) and camera_pose_chain.npz (the composed chain as
the reference's own Camera METHODS compute it, incl. global alignment; make_golden.py runs their bodies on a stub self).
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: int) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def projection_matrix(znear: float, zfar: float, fovX, fovY, device=None) -> torch.Tensor:
    """4x4 OpenGL-style projection P (column-vector form); the op consumes P^T (``intrinsic``)."""
    tan_y = torch.tan(fovY / 2) if torch.is_tensor(fovY) else math.tan(fovY / 2)
    tan_x = torch.tan(fovX / 2) if torch.is_tensor(fovX) else math.tan(fovX / 2)
    top = tan_y * znear
    right = tan_x * znear
    bottom, left = -top, -right
    if device is None and torch.is_tensor(fovX):
        device = fovX.device
    rows = [[None] * 4 for _ in range(4)]
    zero = torch.zeros((), device=device)
    ent = {
        (0, 0): 2.0 * znear / (right - left),
        (1, 1): 2.0 * znear / (top - bottom),
        (0, 2): (right + left) / (right - left),
        (1, 2): (top + bottom) / (top - bottom),
        (3, 2): 1.0,
        (2, 2): zfar / (zfar - znear),
        (2, 3): -(zfar * znear) / (zfar - znear),
    }
    for r in range(4):
        for c in range(4):
            e = ent.get((r, c), 0.0)
            rows[r][c] = e.to(torch.float32) if torch.is_tensor(e) else zero + float(e)
    return torch.stack([torch.stack(r) for r in rows])


def quaternion_to_rotation(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> 3x3, normalising first."""
    q = q / torch.norm(q)
    w, x, y, z = q.unbind(-1)
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz, wx, wy, wz = x * y, x * z, y * z, w * x, w * y, w * z
    return torch.stack([
        torch.stack([1 - 2 * y2 - 2 * z2, 2 * xy - 2 * wz, 2 * xz + 2 * wy]),
        torch.stack([2 * xy + 2 * wz, 1 - 2 * x2 - 2 * z2, 2 * yz - 2 * wx]),
        torch.stack([2 * xz - 2 * wy, 2 * yz + 2 * wx, 1 - 2 * x2 - 2 * y2]),
    ])


def rotation_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """3x3 -> (w,x,y,z)   (scene/cameras.py:418-447 branch structure)."""
    t = R.trace()
    if t > 0:
        s = torch.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = torch.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = torch.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = torch.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    return torch.stack([torch.as_tensor(c, dtype=R.dtype, device=R.device) for c in q])


class PoseCamera(torch.nn.Module):
    """Learnable pinhole camera with the reference's leaves: delta_quaternion (4), delta_translation (3,1),
    learnable_fovx, learnable_fovy  (scene/cameras.py:95-110).  ``R`` is the camera-to-world rotation and
    ``T`` the world-to-camera translation, as COLMAP readers hand them over (scene/dataset_readers.py:401-410)."""

    def __init__(self, R, T, FoVx: float, FoVy: float, width: int, height: int, device="cpu",
                 znear: float = 0.01, zfar: float = 100.0):
        super().__init__()
        dev = torch.device(device)
        self.image_width, self.image_height = int(width), int(height)
        self.FoVx, self.FoVy = float(FoVx), float(FoVy)
        self.znear, self.zfar = znear, zfar
        R = torch.as_tensor(R, dtype=torch.float32)
        self.register_buffer("init_translation", torch.as_tensor(T, dtype=torch.float32).view(3, 1).to(dev))
        self.register_buffer("init_quaternion", rotation_to_quaternion(R.t()).to(dev))
        self.register_buffer("last_row", torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev))
        self.delta_translation = torch.nn.Parameter(torch.zeros(3, 1, device=dev))
        self.delta_quaternion = torch.nn.Parameter(torch.zeros(4, device=dev))
        self.learnable_fovx = torch.nn.Parameter(torch.tensor(self.FoVx, device=dev))
        self.learnable_fovy = torch.nn.Parameter(torch.tensor(self.FoVy, device=dev))

    def get_world_view_transform(self, global_rotation: Optional[torch.Tensor] = None,
                                 global_translation_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        q = self.init_quaternion + self.delta_quaternion
        rot = quaternion_to_rotation(q)
        if global_rotation is not None:
            rot = global_rotation @ rot
        t = self.init_translation + self.delta_translation
        w2c_t = torch.cat((torch.cat((rot, t), dim=1), self.last_row), dim=0).t()
        if global_translation_scale is not None:
            c2w = w2c_t.inverse()
            mask = torch.ones_like(c2w)
            mask[3, :3] = global_translation_scale
            w2c_t = (c2w * mask).inverse()
        return w2c_t

    def get_intrinsic(self) -> torch.Tensor:
        return projection_matrix(self.znear, self.zfar, self.learnable_fovx, self.learnable_fovy).transpose(0, 1)

    def get_full_proj_transform(self, global_rotation=None, global_translation_scale=None) -> torch.Tensor:
        v = self.get_world_view_transform(global_rotation, global_translation_scale)
        return (v.unsqueeze(0).bmm(self.get_intrinsic().unsqueeze(0))).squeeze(0)

    def get_camera_center(self, global_rotation=None, global_translation_scale=None) -> torch.Tensor:
        return self.get_world_view_transform(global_rotation, global_translation_scale).inverse()[3, :3]

    def pose_leaves(self):
        return [self.delta_quaternion, self.delta_translation, self.learnable_fovx, self.learnable_fovy]

    def get_matrices(self, global_rotation: Optional[torch.Tensor] = None,
                     global_translation_scale: Optional[torch.Tensor] = None):
        """(viewmatrix, projmatrix, intrinsic, campos) for GaussianRasterizationSettings.  On a GPU this is ONE HIP launch
        (and one for the backward) instead of the ~40 PyTorch kernels of the four getters above; on the CPU it is those
        getters (each evaluated once)."""
        if self.delta_quaternion.is_cuda:
            return fused_camera_chain(self.delta_quaternion, self.delta_translation, self.learnable_fovx, self.learnable_fovy,
                                      self.init_quaternion, self.init_translation, self.znear, self.zfar,
                                      global_rotation, global_translation_scale)
        v = self.get_world_view_transform(global_rotation, global_translation_scale)
        k = self.get_intrinsic()
        return v, (v.unsqueeze(0).bmm(k.unsqueeze(0))).squeeze(0), k, v.inverse()[3, :3]


class _FusedCameraChain(torch.autograd.Function):
    """Pose leaves -> the four camera tensors through bags_camera_forward / bags_camera_backward (csrc/camera.hip)."""

    @staticmethod
    def forward(ctx, dq, dt, fovx, fovy, q0, t0, znear, zfar, grot, gscale):
        from . import _lib as L
        dev = dq.device
        if not dq.is_cuda:
            raise RuntimeError("fused_camera_chain: tensors must live on a GPU (use PoseCamera's getters on the host)")

        def f32(t, n):
            if t is None:
                return None
            t = t.detach().to(dev, torch.float32).contiguous().reshape(-1)
            if t.numel() != n:
                raise RuntimeError(f"fused_camera_chain: expected {n} values, got {t.numel()}")
            return t
        keep = dict(q0=f32(q0, 4), dq=f32(dq, 4), t0=f32(t0, 3), dt=f32(dt, 3), fovx=f32(fovx, 1), fovy=f32(fovy, 1),
                    grot=f32(grot, 9), gscale=f32(gscale, 1))
        p = lambda t: None if t is None else t.data_ptr()
        cam = L.BagsCamera(p(keep["q0"]), p(keep["dq"]), p(keep["t0"]), p(keep["dt"]), p(keep["fovx"]), p(keep["fovy"]),
                           p(keep["grot"]), p(keep["gscale"]), float(znear), float(zfar))
        out = torch.empty(51, dtype=torch.float32, device=dev)
        V, M, K, Cc = out[0:16].view(4, 4), out[16:32].view(4, 4), out[32:48].view(4, 4), out[48:51]
        lib = L.load()
        with torch.cuda.device(dev):
            L.check(lib.bags_camera_forward(cam, V.data_ptr(), M.data_ptr(), K.data_ptr(), Cc.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "bags_camera_forward")
        ctx.keep, ctx.cam = keep, cam
        ctx.shapes = (dq.shape, dt.shape, fovx.shape, fovy.shape, None if grot is None else grot.shape,
                      None if gscale is None else gscale.shape)
        return V, M, K, Cc

    @staticmethod
    def backward(ctx, gV, gM, gK, gC):
        from . import _lib as L
        keep, cam = ctx.keep, ctx.cam
        dev = keep["dq"].device
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        gV, gM, gK, gC = c(gV), c(gM), c(gK), c(gC)
        need = ctx.needs_input_grad
        out = torch.empty(19, dtype=torch.float32, device=dev)                  # dq 4 | dt 3 | fovx | fovy | grot 9 | gscale (every requested
        # slice is fully written by the kernel, the others are not returned: no fill launch)
        g_dq, g_dt, g_fx, g_fy, g_gr, g_gs = out[0:4], out[4:7], out[7:8], out[8:9], out[9:18], out[18:19]
        p = lambda t, on=True: None if (t is None or not on) else t.data_ptr()
        lib = L.load()
        with torch.cuda.device(dev):
            L.check(lib.bags_camera_backward(cam, p(gV), p(gM), p(gK), p(gC), p(g_dq, need[0]), p(g_dt, need[1]),
                                             p(g_fx, need[2]), p(g_fy, need[3]), p(g_gr, need[8] and keep["grot"] is not None),
                                             p(g_gs, need[9] and keep["gscale"] is not None),
                                             torch.cuda.current_stream().cuda_stream), "bags_camera_backward")
        sh = ctx.shapes
        r = lambda g, on, shape: g.reshape(shape) if (on and shape is not None) else None
        return (r(g_dq, need[0], sh[0]), r(g_dt, need[1], sh[1]), r(g_fx, need[2], sh[2]), r(g_fy, need[3], sh[3]), None, None,
                None, None, r(g_gr, need[8], sh[4]), r(g_gs, need[9], sh[5]))


def fused_camera_chain(delta_quaternion, delta_translation, fovx, fovy, init_quaternion, init_translation,
                       znear: float = 0.01, zfar: float = 100.0, global_rotation=None, global_translation_scale=None):
    """(viewmatrix, projmatrix, intrinsic, campos), differentiable w.r.t. the four pose leaves (and the global alignment
    when given).  Mirrors scene/cameras.py:356-381."""
    return _FusedCameraChain.apply(delta_quaternion, delta_translation, fovx, fovy, init_quaternion, init_translation,
                                   znear, zfar, global_rotation, global_translation_scale)
