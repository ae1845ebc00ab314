"""CPU oracle for the pose-differentiable Gaussian rasterizer  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path (``bags_raster``) never touches it
and fails loudly when the HIP library is missing.

PARITY UNPINNED against the CUDA fork: the rasterizer the reference calls
(``diff_gaussian_rasterization`` = denghilbert/3dgs-pose @ cd77ced15a278bd1e9c0e80c24d61de3a6fe1f3b,
named only at /root/reference/README.md:126) is an empty git submodule, so there is no source, golden
vector or test of the reference to check this restatement against.  What IS pinned (tests/golden):
the SH basis (utils/sh_utils.py:57-112), the projection matrix (utils/graphics_utils.py:83-107), the
quaternion->rotation map (scene/cameras.py:399-416) and the L1/SSIM loss (utils/loss_utils.py).  The
algorithm itself follows the published 3DGS tile rasterizer with the call
contract of gaussian_renderer/__init__.py:30-133.

Design of the oracle
--------------------
* plain PyTorch on CPU, every gradient produced by autograd (nothing hand-derived here, so the
  HIP backward is checked against an independent derivation);
* ``dtype=torch.float32`` mode mirrors the kernels' index-producing arithmetic op for op (explicit
  scalar expressions, left-to-right, no matmul) so radii / tile rects / depth keys / sorted lists are
  comparable BIT-EXACTLY;  ``dtype=torch.float64`` mode is the gradient reference and may be handed the
  fp32 run's discrete decisions (``discrete=``): it then walks the same instance lists AND takes every
  per-pair threshold decision from an fp32 evaluation, so it differs from an fp32 implementation by
  arithmetic only;
* blending is evaluated tile by tile as dense (256 x N) tensors; backward runs tile by tile as well
  (bounded memory), accumulating into per-Gaussian 2-D leaves that are then pulled back through the
  preprocess graph;
* the dense formulation is held against the published per-pixel loops and their hand-derived back-to-front
  recurrences (oracle/published_blend.py, tests/test_oracle_cpu.py): identical n_contrib, values and 2-D
  gradients to float64 round-off; the per-Gaussian backward (incl. D8 and the four camera tensors) likewise
  against the hand-derived chain of oracle/published_preprocess.py.

Semantics decided here because the fork is unavailable (also listed in DESIGN.md):
  D1  focal lengths come from ``intrinsic``: fx = intrinsic[0,0]*W/2, fy = intrinsic[1,1]*H/2
      (equal to W/(2 tanfovx) when the learnable fov equals the static one, scene/cameras.py:109-111);
      the frustum clamp keeps the static tanfovx/tanfovy of the settings.
  D2  shift_factors f: theta = angle(p_view, +z); s = f0 th^3 + f1 th^5 + f2 th^7 (train.py:210-222);
      p_view.z += s and p_hom += s * intrinsic[2,:].  Identity at f = 0 (the only reference behaviour).
      theta comes from a libm-free atan (same operation sequence as the kernels) so indices stay bit-exact for f != 0.
  D3  alpha = min(0.99, o*G) back-propagates straight through the clamp (stock CUDA behaviour).
  D4  means2D is an additive NDC offset (zeros) => its gradient is in NDC units (x W/2 of pixel units),
      means2D_densify receives sum over pixels of |per-pixel NDC gradient| (abs-grad densification,
      scene/gaussian_model.py:449-452).
  D5  extra outputs: depth (1,H,W) = sum w_i z_i, weights (1,H,W) = 1 - T_final, mean2D (P,2) pixel
      centres; depth/weights/mean2D carry no gradient.
  D6  depth key = view-space z (README.md:126 default) or Euclidean distance (``depth_key='distance'``).
  D8  frustum clamp: for a point outside 1.3 x the field of view the EWA Jacobian uses t.x = +-1.3 tanfovx * t.z.  Default
      ``clamp_grad="stock"`` (since round 4, here and in preprocess_bwd.hip): upstream 3DGS's rule, which the reference's fork
      inherits -- dL/dt.x is zeroed (x_grad_mul) and the clamped t.x is a CONSTANT inside dL/dt.z (2 h_x t.x / t.z^3 dL/dJ02).
      ``clamp_grad="exact"`` differentiates the clamped expression itself (dL/dt.z sees t.x move with t.z: half of that one
      term).  The two differ on clamped Gaussians only (tests/test_parity_gpu.py::test_frustum_clamp_gradient_semantic);
      forward values are identical.
  D9  conic = cov2D^-1: default ``conic_grad="stock"`` (round 5, here and in preprocess_bwd.hip): upstream computeCov2DCUDA's
      backward, which multiplies every term by 1 / (det^2 + 1e-7) where the derivative of the inverse has 1 / det^2
      (_ConicStock below restates its three lines); ``conic_grad="exact"`` lets autograd differentiate the division.
      det >= 0.09, so the two differ by <= 1.2e-5 relative on dL/dcov2D; forward values are identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def eval_sh_rgb(deg: int, sh: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """SH -> RGB before the +0.5/clamp.  sh (P,M,3) coefficient-major, d (P,3) unit directions.
    Basis and signs follow utils/sh_utils.py:57-112 (pinned by tests/golden/sh_basis.npz)."""
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                 + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                 + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                     + SH_C3[1] * xy * z * sh[:, 10]
                     + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                     + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                     + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                     + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                     + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return r


@dataclass
class OracleSettings:
    """Mirror of GaussianRasterizationSettings (gaussian_renderer/__init__.py:50-65)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    intrinsic: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False
    debug_iter: Optional[int] = None
    depth_key: str = "z"
    tile_bounds: str = "opacity"      # "aabb": stock 3-sigma square; "opacity": intersected with the alpha >= 1/255 bounds
    clamp_grad: str = "stock"         # D8; "stock": upstream's x_grad_mul treatment of frustum-clamped points; "exact"
    conic_grad: str = "stock"         # D9; "stock": upstream's det^2 + 1e-7 in the backward of the 2x2 inverse; "exact"


@dataclass
class Preprocessed:
    # differentiable per-Gaussian 2-D state
    xy: torch.Tensor          # (P,2) pixel centre
    conic: torch.Tensor       # (P,3)
    opacity: torch.Tensor     # (P,)
    rgb: torch.Tensor         # (P,3)
    depth: torch.Tensor       # (P,) sort key value (detached use only)
    # discrete artefacts
    radii: torch.Tensor       # (P,) int32
    rect: torch.Tensor        # (P,4) int32  minx, miny, maxx, maxy (max exclusive)
    tiles_touched: torch.Tensor  # (P,) int32  instances emitted for the Gaussian (tiles of `rect` that `keep` retains)
    keep: torch.Tensor        # (P,) int64  bit ry * 8 + rx: tile (rect.minx + rx, rect.miny + ry) is emitted; only read for
                              #             rectangles of at most 8 x 8 tiles (larger ones emit every tile)
    clamped: torch.Tensor     # (P,3) bool
    visible: torch.Tensor     # (P,) bool
    extras: Dict[str, torch.Tensor] = field(default_factory=dict)


def _f(v, like):
    return torch.as_tensor(v, dtype=like.dtype)


def _sqrt(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded sqrt.  torch's vectorised fp32 CPU sqrt is NOT correctly rounded (0.6 % of inputs are 1 ulp
    off numpy / the GPU's IEEE sqrt), which would break bit-exact depth keys and radii; sqrt in fp64 rounded once to
    fp32 is exact (53 >= 2*24+2 bits makes the double rounding innocuous)."""
    if x.dtype == torch.float32:
        return torch.sqrt(x.double()).float()
    return torch.sqrt(x)


def _atan2_pos(rho: torch.Tensor, tz: torch.Tensor) -> torch.Tensor:
    """atan2(rho, tz) for rho >= 0, tz > 0.  fp32: the kernels' libm-free sequence (csrc/bags_common.h: det_atan2_pos), op
    for op, so that depth keys and rectangles stay bit-comparable when shift_factors != 0 (autograd differentiates the
    polynomial: its derivative is 1/(1+w^2) to ~1e-6).  fp64: torch.atan2, the gradient reference."""
    if rho.dtype != torch.float32:
        return torch.atan2(rho, tz)
    f = lambda v: _f(v, rho)
    lo, hi = torch.minimum(rho, tz), torch.maximum(rho, tz)
    u = lo / hi
    red = u.detach() > f(0.414213568)
    w = torch.where(red, (u - 1.0) / (u + 1.0), u)
    s = w * w
    p = f(-0.0607120693) * s + f(0.105907366)
    p = p * s + f(-0.142430589)
    p = p * s + f(0.199984416)
    p = p * s + f(-0.333333135)
    a = w + w * (s * p)
    a = torch.where(red, f(0.785398185) + a, a)
    return torch.where(rho.detach() > tz.detach(), f(1.57079637) - a, a)


class _ConicStock(torch.autograd.Function):
    """conic = (cyy, -cxy, cxx) / det with upstream's backward (diff-gaussian-rasterization, computeCov2DCUDA):

        denom2inv = 1 / (det^2 + 1e-7)
        dL/da = denom2inv (-c^2 Gx + 2 b c Gy' + (det - a c) Gz)          a, b, c = cov2D xx, xy, yy
        dL/dc = denom2inv (-a^2 Gz + 2 a b Gy' + (det - a c) Gx)          G = dL/dconic, upstream's Gy' = Gy / 2 (its blend
        dL/db = denom2inv 2 (b c Gx - (det + 2 b^2) Gy' + a b Gz)          backward accumulates -0.5 G dx dy for conic.y)

    The forward values are the exact division, operation for operation what the plain path does (bit-exact integers do not
    depend on the switch).  The sums are formed in float64 and rounded once: the expanded polynomial cancels by lambda1 / lambda2
    on needle-shaped splats ("Needle-shaped splats"), which would make the fp32 oracle the inaccurate side."""

    @staticmethod
    def forward(ctx, cxx, cxy, cyy):
        det = cxx * cyy - cxy * cxy
        ok = det != 0.0
        det_inv = 1.0 / torch.where(ok, det, torch.ones_like(det))
        ctx.save_for_backward(cxx, cxy, cyy)
        return cyy * det_inv, -cxy * det_inv, cxx * det_inv

    @staticmethod
    def backward(ctx, ga, gb, gc):
        a, b, c = (t.double() for t in ctx.saved_tensors)
        Gx, Gy, Gz = ga.double(), 0.5 * gb.double(), gc.double()
        det = a * c - b * b
        d2i = 1.0 / (det * det + 1.0e-7)
        da = d2i * (-c * c * Gx + 2.0 * b * c * Gy + (det - a * c) * Gz)
        dc = d2i * (-a * a * Gz + 2.0 * a * b * Gy + (det - a * c) * Gx)
        db = d2i * 2.0 * (b * c * Gx - (det + 2.0 * b * b) * Gy + a * b * Gz)
        dt = ctx.saved_tensors[0].dtype
        return da.to(dt), db.to(dt), dc.to(dt)


def _rect_full_mask(w, h):
    """Tile mask of a rectangle that emits all its tiles (bags_common.h: rect_full_mask): bit ry * 8 + rx for rx < w, ry < h
    when w, h <= 8; all ones for a larger rectangle; 0 for an empty one."""
    small = (w <= 8) & (h <= 8)
    ws, hs = torch.where(small, w, torch.ones_like(w)).clamp(min=1), torch.where(small, h, torch.ones_like(h)).clamp(min=1)
    rows = torch.full_like(w, 0x0101010101010101) >> (8 * (8 - hs))
    m = ((torch.ones_like(w) << ws) - 1) * rows
    m = torch.where(small, m, torch.full_like(m, -1))
    return torch.where((w <= 0) | (h <= 0), torch.zeros_like(m), m)


def _tile_reach(px, py, a, b, c, tau2m, minx, miny, w, h):
    """Which tiles of a small rectangle (w, h <= 8) the ellipse Q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau2m around (px, py)
    reaches: the minimum of Q over the square of a tile's pixel centres [16 tx, 16 tx + 15] x [16 ty, 16 ty + 15] lies, for a
    centre outside the square, on an edge that faces the centre (Q is convex with its minimum at the centre), and along
    an edge at the clamped 1-D minimiser.  Same operations in the same order as tile_reach in preprocess_fwd.hip (adds,
    multiplies, two reciprocals per Gaussian, min / max: correctly rounded in both), so the masks are bit-identical.
    Returns (bits int64 -- bit ry * 8 + rx, count int64)."""
    n = px.shape[0]
    bits = torch.zeros(n, dtype=torch.int64)
    cnt = torch.zeros(n, dtype=torch.int64)
    ra, rc, b2 = 1.0 / a, 1.0 / c, 2.0 * b
    inf = torch.full_like(px, float("inf"))
    for ry in range(int(h.max())):
        Y0 = ((miny + ry) * TILE).to(px.dtype); Y1 = Y0 + 15.0
        yin = (py >= Y0) & (py <= Y1)
        dyE = torch.where(py < Y0, Y0, Y1) - py
        xs = px - (b * dyE) * ra
        for rx in range(int(w.max())):
            live = (ry < h) & (rx < w)
            if not bool(live.any()):
                continue
            X0 = ((minx + rx) * TILE).to(px.dtype); X1 = X0 + 15.0
            xin = (px >= X0) & (px <= X1)
            dx = torch.minimum(torch.maximum(xs, X0), X1) - px
            qh = (a * dx) * dx + (b2 * dx) * dyE + (c * dyE) * dyE
            dxE = torch.where(px < X0, X0, X1) - px
            ys = py - (b * dxE) * rc
            dy = torch.minimum(torch.maximum(ys, Y0), Y1) - py
            qv = (a * dxE) * dxE + (b2 * dxE) * dy + (c * dy) * dy
            q = torch.minimum(torch.where(yin, inf, qh), torch.where(xin, inf, qv))
            k = live & ((xin & yin) | ~(q > tau2m))
            bits = bits | (k.to(torch.int64) << (ry * 8 + rx))
            cnt = cnt + k.to(torch.int64)
    return bits, cnt


def preprocess(means3D, means2D, shift_factors, shs, colors_precomp, opacities, scales, rotations,
               cov3D_precomp, s: OracleSettings, dtype=torch.float32,
               discrete: Optional[Dict[str, torch.Tensor]] = None) -> Preprocessed:
    """Per-Gaussian projection, EWA covariance, radius, tile rectangle, colour  (SURVEY Appendix A.1).

    Every expression below is the exact fp32 operation sequence of the HIP kernel ``preprocess_fwd``
    (csrc/bags_raster.hip): a*b + c*d is two rounded products and one rounded sum, evaluated left to right.
    The differentiable graph is built on the near-plane survivors only, so culled Gaussians can never leak a
    0*inf into the pose-gradient sums.
    """
    P = means3D.shape[0]
    W, H = int(s.image_width), int(s.image_height)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    c = lambda t: None if t is None else t.to(dtype)
    means3D, means2D, opacities = c(means3D), c(means2D), c(opacities)
    shs, colors_precomp, scales, rotations, cov3D_precomp = c(shs), c(colors_precomp), c(scales), c(rotations), c(cov3D_precomp)
    v = c(s.viewmatrix).reshape(16)
    m = c(s.projmatrix).reshape(16)
    k = c(s.intrinsic).reshape(16)
    campos = c(s.campos).reshape(3)
    sf = c(shift_factors) if shift_factors is not None else torch.zeros(3, dtype=dtype)

    with torch.no_grad():
        if discrete is None:
            tz_all = means3D[:, 0] * v[2] + means3D[:, 1] * v[6] + means3D[:, 2] * v[10] + v[14]
            cand = tz_all > 0.2                      # near-plane cull (stock in_frustum)
        else:
            cand = discrete["radii"] > 0
        idx = cand.nonzero().squeeze(1)
    n = idx.numel()
    sub = lambda t: None if t is None else t.index_select(0, idx)
    m3, m2, op_s = sub(means3D), sub(means2D), sub(opacities)
    shs_s, col_s, sc_s, rot_s, cov_s = sub(shs), sub(colors_precomp), sub(scales), sub(rotations), sub(cov3D_precomp)

    x, y, z = m3[:, 0], m3[:, 1], m3[:, 2]
    # p_view = [x y z 1] . viewmatrix   (row-vector convention, utils/graphics_utils.py:26-33)
    tx = x * v[0] + y * v[4] + z * v[8] + v[12]
    ty = x * v[1] + y * v[5] + z * v[9] + v[13]
    tz = x * v[2] + y * v[6] + z * v[10] + v[14]

    # D2: entrance-pupil shift along the optical axis (identity when shift_factors == 0)
    rho = _sqrt(tx * tx + ty * ty + 1e-20)
    theta = _atan2_pos(rho, tz)
    th2 = theta * theta
    th3 = th2 * theta
    shift = sf[0] * th3 + sf[1] * (th3 * th2) + sf[2] * (th3 * th2 * th2)
    tzs = tz + shift

    # p_hom = [x y z 1] . projmatrix  (+ shift * intrinsic[2,:])
    hx = x * m[0] + y * m[4] + z * m[8] + m[12] + shift * k[8]
    hy = x * m[1] + y * m[5] + z * m[9] + m[13] + shift * k[9]
    hw = x * m[3] + y * m[7] + z * m[11] + m[15] + shift * k[11]
    pw = 1.0 / (hw + 1e-7)
    ndc_x = hx * pw + m2[:, 0]
    ndc_y = hy * pw + m2[:, 1]
    px = ((ndc_x + 1.0) * W - 1.0) * 0.5
    py = ((ndc_y + 1.0) * H - 1.0) * 0.5

    # 3-D covariance  Sigma = (R S)(R S)^T   (utils/general_utils.py:130-163, scene/gaussian_model.py:37-42)
    if cov_s is not None:
        c0, c1, c2, c3, c4, c5 = [cov_s[:, i] for i in range(6)]
    else:
        mod = float(s.scale_modifier)
        s0, s1, s2 = sc_s[:, 0] * mod, sc_s[:, 1] * mod, sc_s[:, 2] * mod
        qr, qx, qy, qz = rot_s[:, 0], rot_s[:, 1], rot_s[:, 2], rot_s[:, 3]
        r00 = 1.0 - 2.0 * (qy * qy + qz * qz)
        r01 = 2.0 * (qx * qy - qr * qz)
        r02 = 2.0 * (qx * qz + qr * qy)
        r10 = 2.0 * (qx * qy + qr * qz)
        r11 = 1.0 - 2.0 * (qx * qx + qz * qz)
        r12 = 2.0 * (qy * qz - qr * qx)
        r20 = 2.0 * (qx * qz - qr * qy)
        r21 = 2.0 * (qy * qz + qr * qx)
        r22 = 1.0 - 2.0 * (qx * qx + qy * qy)
        l00, l01, l02 = r00 * s0, r01 * s1, r02 * s2
        l10, l11, l12 = r10 * s0, r11 * s1, r12 * s2
        l20, l21, l22 = r20 * s0, r21 * s1, r22 * s2
        c0 = l00 * l00 + l01 * l01 + l02 * l02
        c1 = l00 * l10 + l01 * l11 + l02 * l12
        c2 = l00 * l20 + l01 * l21 + l02 * l22
        c3 = l10 * l10 + l11 * l11 + l12 * l12
        c4 = l10 * l20 + l11 * l21 + l12 * l22
        c5 = l20 * l20 + l21 * l21 + l22 * l22

    # EWA 2-D covariance  cov = J Wc Sigma Wc^T J^T, Wc[j][k] = viewmatrix[k][j]
    fx = k[0] * (0.5 * W)      # D1
    fy = k[5] * (0.5 * H)
    limx = _f(1.3, tx) * _f(s.tanfovx, tx)
    limy = _f(1.3, tx) * _f(s.tanfovy, tx)
    txtz = tx / tzs
    tytz = ty / tzs
    cx_ = torch.minimum(limx, torch.maximum(-limx, txtz)) * tzs
    cy_ = torch.minimum(limy, torch.maximum(-limy, tytz)) * tzs
    if s.clamp_grad == "stock":
        # Upstream 3DGS (computeCov2DCUDA backward): a clamped t.x is a CONSTANT in J (x_grad_mul zeroes dL/dt.x, and dL/dt.z
        # takes 2 h_x t.x / t.z^3 with that constant), although t.x = +-1.3 tanfov * t.z moves with t.z.  Values identical.
        with torch.no_grad():
            clx = (txtz < -limx) | (txtz > limx)
            cly = (tytz < -limy) | (tytz > limy)
        cx_ = torch.where(clx, cx_.detach(), cx_)
        cy_ = torch.where(cly, cy_.detach(), cy_)
    itz = 1.0 / tzs
    itz2 = itz * itz
    j00 = fx * itz
    j02 = -(fx * cx_) * itz2
    j11 = fy * itz
    j12 = -(fy * cy_) * itz2
    a00 = j00 * v[0] + j02 * v[2]
    a01 = j00 * v[4] + j02 * v[6]
    a02 = j00 * v[8] + j02 * v[10]
    a10 = j11 * v[1] + j12 * v[2]
    a11 = j11 * v[5] + j12 * v[6]
    a12 = j11 * v[9] + j12 * v[10]
    b00 = a00 * c0 + a01 * c1 + a02 * c2
    b01 = a00 * c1 + a01 * c3 + a02 * c4
    b02 = a00 * c2 + a01 * c4 + a02 * c5
    b10 = a10 * c0 + a11 * c1 + a12 * c2
    b11 = a10 * c1 + a11 * c3 + a12 * c4
    b12 = a10 * c2 + a11 * c4 + a12 * c5
    cxx = b00 * a00 + b01 * a01 + b02 * a02 + 0.3
    cxy = b00 * a10 + b01 * a11 + b02 * a12
    cyy = b10 * a10 + b11 * a11 + b12 * a12 + 0.3
    det = cxx * cyy - cxy * cxy
    det_ok = det.detach() != 0.0
    if s.conic_grad == "stock":
        con_a, con_b, con_c = _ConicStock.apply(cxx, cxy, cyy)
    else:
        det_inv = 1.0 / torch.where(det_ok, det, torch.ones_like(det))
        con_a = cyy * det_inv
        con_b = -cxy * det_inv
        con_c = cxx * det_inv
    mid = 0.5 * (cxx + cyy)
    lam = mid + _sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius_f = torch.ceil(3.0 * _sqrt(lam))

    # colour
    if col_s is not None:
        rgb = col_s
        clamped_s = torch.zeros(n, 3, dtype=torch.bool)
    else:
        dx_, dy_, dz_ = x - campos[0], y - campos[1], z - campos[2]
        dl = _sqrt(dx_ * dx_ + dy_ * dy_ + dz_ * dz_)
        d = torch.stack([dx_ / dl, dy_ / dl, dz_ / dl], 1)
        raw = eval_sh_rgb(int(s.sh_degree), shs_s, d) + 0.5
        clamped_s = raw.detach() < 0
        rgb = torch.clamp_min(raw, 0.0)

    if s.depth_key == "distance":
        depth = _sqrt(tx * tx + ty * ty + tzs * tzs)
    else:
        depth = tzs

    def full(t, fill=0.0):
        shape = (P,) + tuple(t.shape[1:])
        base = torch.full(shape, fill, dtype=t.dtype)
        return base.index_copy(0, idx, t)

    if discrete is None:
        big = 1.0e9
        def tile_lo(p, r, g):
            q = torch.trunc(torch.clamp((p - r) / TILE, -big, big)).to(torch.int64)
            return torch.clamp(q, 0, g)
        def tile_hi(p, r, g):
            q = torch.trunc(torch.clamp((p + r + (TILE - 1)) / TILE, -big, big)).to(torch.int64)
            return torch.clamp(q, 0, g)
        pxd, pyd, rd = px.detach(), py.detach(), radius_f.detach()
        ok = det_ok & torch.isfinite(pxd) & torch.isfinite(pyd) & torch.isfinite(rd)
        pxd = torch.where(ok, pxd, torch.zeros_like(pxd)); pyd = torch.where(ok, pyd, torch.zeros_like(pyd))
        rd = torch.where(ok, rd, torch.zeros_like(rd))
        minx, maxx = tile_lo(pxd, rd, gx), tile_hi(pxd, rd, gx)
        miny, maxy = tile_lo(pyd, rd, gy), tile_hi(pyd, rd, gy)
        tiles = (maxx - minx) * (maxy - miny)
        vis_s = ok & (tiles > 0)                       # "visible" (radii > 0) is decided by the stock rectangle
        keep_s = torch.full((n,), -1, dtype=torch.int64)
        masked_s = torch.zeros(n, dtype=torch.bool)
        if s.tile_bounds == "opacity":
            # Decision D7 (include/bags_raster.h: BAGS_TILES_OPACITY): emit instances only for tiles inside the axis-aligned
            # bounds of the ellipse alpha >= 1/255.  Same operations, same order as preprocess_fwd.hip (every one of
            # them is correctly rounded in both implementations, so the rectangles are bit-identical).
            f = lambda v: _f(v, pxd)
            visv = f(255.0) * op_s.reshape(n).detach()
            has = ok & (visv >= 1.0)
            visv = torch.where(has, visv, torch.ones_like(visv))
            mant, e2 = torch.frexp(visv)
            t = mant - 1.0
            poly = t * (1.0 + t * (-0.5 + t * (f(0.33333334) + t * -0.25)))
            lnu = e2.to(pxd.dtype) * f(0.6931472) + poly + f(1.0e-3)
            tau2 = 2.0 * lnu + f(0.02)
            rx = _sqrt(tau2 * cxx.detach()) * f(1.02) + f(0.1)
            ry = _sqrt(tau2 * cyy.detach()) * f(1.02) + f(0.1)
            rx = torch.where(has, rx, torch.zeros_like(rx)); ry = torch.where(has, ry, torch.zeros_like(ry))
            def t_lo(p, r, g):
                return torch.clamp(torch.trunc(torch.clamp((p - r) / TILE, -big, big)).to(torch.int64), 0, g)
            def t_hi(p, r, g):
                return torch.clamp(torch.trunc(torch.clamp((p + r) / TILE, -big, big)).to(torch.int64) + 1, 0, g)
            ex0, ex1 = torch.maximum(minx, t_lo(pxd, rx, gx)), torch.minimum(maxx, t_hi(pxd, rx, gx))
            ey0, ey1 = torch.maximum(miny, t_lo(pyd, ry, gy)), torch.minimum(maxy, t_hi(pyd, ry, gy))
            ex0 = torch.where(has, ex0, minx); ey0 = torch.where(has, ey0, miny)      # 255 o < 1: nothing can contribute
            empty = (~has) | (ex1 <= ex0) | (ey1 <= ey0)
            ex1 = torch.where(empty, ex0, ex1); ey1 = torch.where(empty, ey0, ey1)
            minx, maxx, miny, maxy = ex0, ex1, ey0, ey1
            tiles = (maxx - minx) * (maxy - miny)
            # D7, second half: inside a rectangle of at most 8 x 8 tiles, a tile is emitted only if the ellipse
            # d^T Q d <= 1.02 tau2 reaches the square of its pixel centres (_tile_reach; preprocess_fwd.hip: tile_reach).
            ca, cb, cc = con_a.detach(), con_b.detach(), con_c.detach()
            small = (vis_s & has & (tiles > 0) & ((maxx - minx) <= 8) & ((maxy - miny) <= 8)
                     & (ca > 0) & (cc > 0) & ((ca * cc - cb * cb) > 0))
            sel = torch.nonzero(small).reshape(-1)
            if sel.numel():
                bits, cnt = _tile_reach(pxd[sel], pyd[sel], ca[sel], cb[sel], cc[sel], tau2[sel] * f(1.02),
                                        minx[sel], miny[sel], (maxx - minx)[sel], (maxy - miny)[sel])
                keep_s = keep_s.index_copy(0, sel, bits)
                masked_s = small
                tiles = tiles.index_copy(0, sel, cnt)
        keep_s = torch.where(masked_s, keep_s, _rect_full_mask(maxx - minx, maxy - miny))
        keep_s = torch.where(vis_s, keep_s, torch.full_like(keep_s, -1))
        tiles = torch.where(vis_s, tiles, torch.zeros_like(tiles))
        radii_s = torch.where(vis_s, rd, torch.zeros_like(rd)).to(torch.int32)
        rect_s = torch.stack([minx, miny, maxx, maxy], 1).to(torch.int32)
        rect_s = torch.where(vis_s[:, None], rect_s, torch.zeros_like(rect_s))
        radii, rect, tiles_touched = full(radii_s, 0), full(rect_s, 0), full(tiles.to(torch.int32), 0)
        keep = full(keep_s, -1)
        visible = full(vis_s, False)
    else:
        radii, rect, tiles_touched = discrete["radii"], discrete["rect"], discrete["tiles_touched"]
        keep = discrete["keep"] if "keep" in discrete else torch.full((P,), -1, dtype=torch.int64)
        visible = radii > 0

    return Preprocessed(xy=full(torch.stack([px, py], 1)), conic=full(torch.stack([con_a, con_b, con_c], 1)),
                        opacity=full(op_s.reshape(n)), rgb=full(rgb), depth=full(depth), radii=radii, rect=rect,
                        tiles_touched=tiles_touched, keep=keep, clamped=full(clamped_s, False), visible=visible,
                        extras={"cov2d": full(torch.stack([cxx, cxy, cyy], 1)), "tz": full(tzs),
                                "_graph": {"idx": idx, "cov": (cxx, cxy, cyy), "A": (a00, a01, a02, a10, a11, a12),
                                           "sigma": (c0, c1, c2, c3, c4, c5)}})


def bin_and_sort(depth32: torch.Tensor, rect: torch.Tensor, tiles_touched: torch.Tensor, gx: int, gy: int,
                 keep: Optional[torch.Tensor] = None):
    """Instance emission, (tile|depth) keys, stable sort, tile ranges  (SURVEY Appendix A.2).

    depth32: (P,) float32 (the fp32 run's depth: its bit pattern is the low half of the key).
    keep: (P,) int64 tile masks of the rectangles of at most 8 x 8 tiles (Preprocessed.keep); None: every tile.
    Returns keys_sorted (I,) int64, point_list (I,) int32, ranges (T,2) int32, offsets (P,) int64 inclusive scan.
    """
    tt = tiles_touched.to(torch.int64)
    offsets = torch.cumsum(tt, 0)
    I = int(offsets[-1]) if tt.numel() else 0
    T = gx * gy
    if I == 0:
        return (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32),
                torch.zeros(T, 2, dtype=torch.int32), offsets)
    wv = (rect[:, 2] - rect[:, 0]).to(torch.int64)
    hv = (rect[:, 3] - rect[:, 1]).to(torch.int64)
    area = torch.where(tt > 0, wv * hv, torch.zeros_like(tt))       # walk the whole rectangle, y outer, x inner ...
    gid = torch.repeat_interleave(torch.arange(tt.numel(), dtype=torch.int64), area)
    local = torch.arange(gid.numel(), dtype=torch.int64) - (torch.cumsum(area, 0) - area)[gid]
    w = wv[gid]
    ry, rx = local // w, local % w
    if keep is not None:                                            # ... and drop the tiles its mask does not retain
        small = (wv <= 8) & (hv <= 8)
        kept = (~small[gid]) | (((keep[gid] >> (ry * 8 + rx).clamp(max=63)) & 1) != 0)
        gid, ry, rx = gid[kept], ry[kept], rx[kept]
    assert gid.numel() == I, "tiles_touched does not match the rectangles and tile masks"
    tyy = rect[:, 1].to(torch.int64)[gid] + ry
    txx = rect[:, 0].to(torch.int64)[gid] + rx
    tile = tyy * gx + txx
    dbits = depth32.to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    keys = (tile << 32) | dbits[gid]
    keys_sorted, order = torch.sort(keys, stable=True)
    point_list = gid[order].to(torch.int32)
    tile_sorted = keys_sorted >> 32
    counts = torch.bincount(tile_sorted, minlength=T)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], 1).to(torch.int32)
    return keys_sorted, point_list, ranges, offsets


def _blend_tile(ids, xy, conic, opacity, rgb, zdepth, bg, px0, py0, W, H, want_pairs=False, decide32=False):
    """Front-to-back alpha blend of one 16x16 tile as dense (256,N) tensors.

    ``decide32``: take the per-(pixel, splat) threshold decisions (power <= 0, alpha >= 1/255, T < 1e-4 stop) from an
    fp32 evaluation of the same pairs while the values are computed in the tensors' own dtype.  Used by the fp64
    gradient reference so that it walks exactly the pairs an fp32 implementation walks: otherwise an ulp of difference
    in exp() flips borderline pairs and the comparison measures threshold flips, not arithmetic."""
    dt = xy.dtype
    jj, ii = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    pxs = (px0 + ii.reshape(-1)).to(dt)
    pys = (py0 + jj.reshape(-1)).to(dt)
    inside = (pxs < W) & (pys < H)
    gxy, gcon, gop, grgb = xy[ids], conic[ids], opacity[ids], rgb[ids]
    dx = gxy[None, :, 0] - pxs[:, None]
    dy = gxy[None, :, 1] - pys[:, None]
    power = -0.5 * (gcon[None, :, 0] * dx * dx + gcon[None, :, 2] * dy * dy) - gcon[None, :, 1] * dx * dy
    G = torch.exp(power)
    a_raw = gop[None, :] * G
    alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()        # D3 straight-through
    if decide32 and dt != torch.float32:
        with torch.no_grad():
            f = torch.float32
            dx32 = gxy[None, :, 0].to(f) - pxs[:, None].to(f)
            dy32 = gxy[None, :, 1].to(f) - pys[:, None].to(f)
            c32 = gcon.to(f)
            p32 = -0.5 * (c32[None, :, 0] * dx32 * dx32 + c32[None, :, 2] * dy32 * dy32) - c32[None, :, 1] * dx32 * dy32
            a32 = torch.clamp(gop.to(f)[None, :] * torch.exp(p32), max=0.99)
            valid = (p32 <= 0) & (a32 >= 1.0 / 255.0) & inside[:, None]
            T32 = torch.cumprod(1.0 - torch.where(valid, a32, torch.zeros_like(a32)), 1)
            stop = valid & (T32 < 1e-4)
    else:
        valid = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0) & inside[:, None]
        stop = None
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    Tincl = torch.cumprod(one_m, 1)                                         # T after splat i
    Texcl = torch.cat([torch.ones(256, 1, dtype=dt), Tincl[:, :-1]], 1)     # T before splat i
    if stop is None:
        stop = valid & (Tincl.detach() < 1e-4)
    live = torch.cumsum(stop.to(torch.int32), 1) == 0                       # the stopping splat is excluded
    contrib = valid & live
    w = torch.where(contrib, a_eff * Texcl, torch.zeros_like(a_eff))
    color = w @ grgb                                                        # (256,3)
    T_final = torch.prod(torch.where(contrib, one_m, torch.ones_like(one_m)), 1)
    out = color + T_final[:, None] * bg[None, :]
    n = ids.numel()
    idx1 = torch.arange(1, n + 1)[None, :].expand(256, n)
    n_contrib = torch.where(contrib, idx1, torch.zeros_like(idx1)).amax(1) if n else torch.zeros(256, dtype=torch.int64)
    dep = (w.detach() @ zdepth[ids].detach().to(dt))
    res = dict(out=out, T_final=T_final, n_contrib=n_contrib, depth=dep, inside=inside)
    if want_pairs:
        res.update(G=G, dx=dx, dy=dy, contrib=contrib, gcon=gcon, gop=gop)
    return res


class OracleState:
    pass


def rasterize_forward(means3D, means2D, shift_factors, shs, colors_precomp, opacities, scales, rotations,
                      cov3D_precomp, s: OracleSettings, dtype=torch.float32, discrete=None,
                      tiles: Optional[torch.Tensor] = None) -> OracleState:
    """Full forward.  Inputs that require grad stay attached (preprocess graph is kept for backward).
    ``tiles``: optional subset of tile ids to blend (bounded CPU-baseline sample)."""
    W, H = int(s.image_width), int(s.image_height)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pre = preprocess(means3D, means2D, shift_factors, shs, colors_precomp, opacities, scales, rotations,
                     cov3D_precomp, s, dtype, discrete)
    if discrete is not None and "point_list" in discrete:
        keys_sorted, point_list, ranges = discrete["keys_sorted"], discrete["point_list"], discrete["ranges"]
        offsets = torch.cumsum(pre.tiles_touched.to(torch.int64), 0)
    else:
        d32 = discrete["depth32"] if discrete is not None else pre.depth.detach().to(torch.float32)
        keys_sorted, point_list, ranges, offsets = bin_and_sort(d32, pre.rect, pre.tiles_touched, gx, gy, pre.keep)
    st = OracleState()
    st.pre, st.s, st.dtype = pre, s, dtype
    st.decide32 = discrete is not None and dtype != torch.float32
    st.keys_sorted, st.point_list, st.ranges, st.offsets = keys_sorted, point_list, ranges, offsets
    st.gx, st.gy, st.W, st.H = gx, gy, W, H
    bg = s.bg.to(dtype).reshape(3)
    image = torch.zeros(3, H, W, dtype=dtype)
    depth_img = torch.zeros(1, H, W, dtype=dtype)
    weights = torch.zeros(1, H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    final_T = torch.ones(H, W, dtype=dtype)
    pl = point_list.to(torch.int64)
    tile_ids = range(gx * gy) if tiles is None else [int(t) for t in tiles]
    with torch.no_grad():
        xy, conic, op, rgb, zd = pre.xy.detach(), pre.conic.detach(), pre.opacity.detach(), pre.rgb.detach(), pre.extras["tz"].detach()
        for t in tile_ids:
            ty, tx = divmod(t, gx)
            ids = pl[int(ranges[t, 0]):int(ranges[t, 1])]
            r = _blend_tile(ids, xy, conic, op, rgb, zd, bg, tx * TILE, ty * TILE, W, H, decide32=st.decide32)
            y0, x0 = ty * TILE, tx * TILE
            hh, ww = min(TILE, H - y0), min(TILE, W - x0)
            image[:, y0:y0 + hh, x0:x0 + ww] = r["out"].reshape(TILE, TILE, 3)[:hh, :ww].permute(2, 0, 1)
            depth_img[0, y0:y0 + hh, x0:x0 + ww] = r["depth"].reshape(TILE, TILE)[:hh, :ww]
            final_T[y0:y0 + hh, x0:x0 + ww] = r["T_final"].reshape(TILE, TILE)[:hh, :ww]
            n_contrib[y0:y0 + hh, x0:x0 + ww] = r["n_contrib"].reshape(TILE, TILE)[:hh, :ww].to(torch.int32)
    weights[0] = 1.0 - final_T
    st.image, st.depth_img, st.weights, st.n_contrib, st.final_T = image, depth_img, weights, n_contrib, final_T
    st.radii = pre.radii
    st.mean2D = torch.where(pre.visible[:, None], pre.xy.detach(), torch.zeros_like(pre.xy.detach()))
    st.tile_ids = tile_ids
    return st


def rasterize_backward(st: OracleState, grad_image: torch.Tensor, inputs: Dict[str, torch.Tensor]):
    """Pull ``grad_image`` (3,H,W) back to every tensor in ``inputs`` (name -> tensor with requires_grad that
    took part in the forward).  Also returns the abs-grad densification accumulator (D4)."""
    pre, s, dtype = st.pre, st.s, st.dtype
    W, H, gx = st.W, st.H, st.gx
    bg = s.bg.to(dtype).reshape(3)
    gimg = grad_image.to(dtype)
    P = pre.xy.shape[0]
    leaves = [t.detach().clone().requires_grad_(True) for t in (pre.xy, pre.conic, pre.opacity, pre.rgb)]
    lxy, lcon, lop, lrgb = leaves
    acc = [torch.zeros_like(t) for t in leaves]
    absgrad = torch.zeros(P, 2, dtype=dtype)
    pl = st.point_list.to(torch.int64)
    zd = pre.extras["tz"].detach()
    for t in st.tile_ids:
        ty, tx = divmod(t, gx)
        lo, hi = int(st.ranges[t, 0]), int(st.ranges[t, 1])
        if hi <= lo:
            continue
        ids = pl[lo:hi]
        r = _blend_tile(ids, lxy, lcon, lop, lrgb, zd, bg, tx * TILE, ty * TILE, W, H, want_pairs=True,
                        decide32=st.decide32)
        y0, x0 = ty * TILE, tx * TILE
        hh, ww = min(TILE, H - y0), min(TILE, W - x0)
        g = torch.zeros(TILE, TILE, 3, dtype=dtype)
        g[:hh, :ww] = gimg[:, y0:y0 + hh, x0:x0 + ww].permute(1, 2, 0)
        loss = (r["out"] * g.reshape(256, 3)).sum()
        grads = torch.autograd.grad(loss, leaves + [r["G"]], allow_unused=True)
        for a, gr in zip(acc, grads[:4]):
            if gr is not None:
                a += gr
        dG = grads[4]
        if dG is not None:
            q = dG * r["G"].detach()
            cxn = r["gcon"].detach()
            dxx, dyy = r["dx"].detach(), r["dy"].detach()
            gx_pix = q * (-(cxn[None, :, 0] * dxx) - cxn[None, :, 1] * dyy)
            gy_pix = q * (-(cxn[None, :, 2] * dyy) - cxn[None, :, 1] * dxx)
            absgrad.index_add_(0, ids, torch.stack([(gx_pix * (0.5 * W)).abs().sum(0),
                                                     (gy_pix * (0.5 * H)).abs().sum(0)], 1))
    # invisible Gaussians never enter a list, so their 2-D leaves got nothing
    names = [n for n, t in inputs.items() if t is not None and t.requires_grad]
    tensors = [inputs[n] for n in names]
    outs = [pre.xy, pre.conic, pre.opacity, pre.rgb]
    keep = [(o, a) for o, a in zip(outs, acc) if o.requires_grad]
    res = {}
    if tensors and keep:
        g = torch.autograd.grad([o for o, _ in keep], tensors, [a for _, a in keep], allow_unused=True)
        for n, t, gg in zip(names, tensors, g):
            res[n] = torch.zeros_like(t) if gg is None else gg
    res["means2D_densify"] = torch.cat([absgrad, torch.zeros(P, 1, dtype=dtype)], 1)
    res["_2d"] = dict(xy=acc[0], conic=acc[1], opacity=acc[2], rgb=acc[3])
    return res


INPUT_NAMES = ("means3D", "means2D", "shift_factors", "shs", "colors_precomp", "opacities", "scales",
               "rotations", "cov3D_precomp", "viewmatrix", "projmatrix", "intrinsic", "campos")


def render_and_grad(inputs: Dict[str, Optional[torch.Tensor]], s: OracleSettings, grad_image: Optional[torch.Tensor],
                    dtype=torch.float32, discrete=None, tiles=None):
    """Convenience: forward (+ backward when ``grad_image`` is given) on detached copies of ``inputs``.
    ``inputs`` holds the 9 call tensors plus viewmatrix/projmatrix/intrinsic/campos (which override ``s``)."""
    leaf = {}
    for n in INPUT_NAMES:
        t = inputs.get(n)
        if t is None:
            leaf[n] = None
        else:
            tt = t.detach().to(dtype).clone()
            leaf[n] = tt.requires_grad_(grad_image is not None)
    s2 = OracleSettings(**{**s.__dict__})
    for n in ("viewmatrix", "projmatrix", "intrinsic", "campos"):
        if leaf[n] is not None:
            setattr(s2, n, leaf[n])
        else:
            leaf[n] = getattr(s, n).detach().to(dtype).clone().requires_grad_(grad_image is not None)
            setattr(s2, n, leaf[n])
    if leaf["means2D"] is None:
        leaf["means2D"] = torch.zeros(leaf["means3D"].shape[0], 3, dtype=dtype, requires_grad=grad_image is not None)
    st = rasterize_forward(leaf["means3D"], leaf["means2D"], leaf["shift_factors"], leaf["shs"], leaf["colors_precomp"],
                           leaf["opacities"], leaf["scales"], leaf["rotations"], leaf["cov3D_precomp"], s2, dtype,
                           discrete, tiles)
    grads = None
    if grad_image is not None:
        grads = rasterize_backward(st, grad_image, leaf)
    return st, grads


def discrete_of(st: OracleState) -> Dict[str, torch.Tensor]:
    """Discrete decisions of an fp32 run, to be replayed by the fp64 gradient reference."""
    return dict(radii=st.pre.radii, rect=st.pre.rect, tiles_touched=st.pre.tiles_touched, keep=st.pre.keep,
                keys_sorted=st.keys_sorted, point_list=st.point_list, ranges=st.ranges,
                depth32=st.pre.depth.detach().to(torch.float32))
