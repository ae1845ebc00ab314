"""Hand-derived backward of the per-Gaussian preprocess, in the structure of the published 3DGS rasterizer  --  TEST
INFRASTRUCTURE ONLY (same rule as raster_oracle.py: tests import it, the product never does).

``raster_oracle.preprocess`` leaves its backward to autograd; decision D8 (the gradient at the frustum clamp) is expressed there
as a ``detach`` of the clamped coordinate.  This file writes the same backward out by hand, the way the published kernels are
organised ([UPSTREAM-KNOWLEDGE]; the reference's own copy is an empty submodule, README.md:126), so that
the oracle's Gaussian-side gradients -- and the ``detach`` reading of D8 in particular -- are held against explicit formulas:

  conic -> cov2D      inverse of the symmetric 2x2 (a b; b c), det = ac - b^2:
                      dL/da = (-c^2 gA + bc gB - b^2 gC) / det^2,  dL/dc = (-a^2 gC + ab gB - b^2 gA) / det^2,
                      dL/db = (2bc gA - (det + 2b^2) gB + 2ab gC) / det^2          (gB = derivative w.r.t. the conic's b itself)
  cov2D -> cov3D, T   cov2D = T Sigma T^T, T = J W (2x3):  dL/dSigma_ii = T0i^2 dL/da + T0i T1i dL/db + T1i^2 dL/dc,
                      dL/dSigma_ij (both symmetric entries) = 2 T0i T0j dL/da + (T0i T1j + T0j T1i) dL/db + 2 T1i T1j dL/dc,
                      dL/dT0 = 2 (Sigma T0) dL/da + (Sigma T1) dL/db,  dL/dT1 = 2 (Sigma T1) dL/dc + (Sigma T0) dL/db
  T -> J -> t         J00 = fx / tz, J02 = -fx tx' / tz^2, J11 = fy / tz, J12 = -fy ty' / tz^2 with the CLAMPED tx', ty';
                      dL/dtx = x_grad_mul (-fx / tz^2) dL/dJ02,  dL/dty = y_grad_mul (-fy / tz^2) dL/dJ12,
                      dL/dtz = -fx / tz^2 dL/dJ00 - fy / tz^2 dL/dJ11 + 2 fx tx' / tz^3 dL/dJ02 + 2 fy ty' / tz^3 dL/dJ12
                      x_grad_mul = 0 where |tx / tz| > 1.3 tanfovx (else 1): the published rule, decision D8 "stock".
                      clamp_grad="exact" instead differentiates tx' = +-lim tz: dL/dtz takes only ONE fx tx' / tz^3 dL/dJ02.
  t -> mean           dL/dmean_k += R[k][0] dL/dtx + R[k][1] dL/dty + R[k][2] dL/dtz   (R[k][c] = viewmatrix[k][c])
  pixel -> mean       p_hom = [mean 1] . projmatrix, m_w = 1 / (p_hom.w + 1e-7), ndc = p_hom.xy m_w, pixel = ((ndc + 1) W - 1) / 2
  colour              rgb = max(0, sum_t basis_t(dir) sh_t + 0.5): dL/dsh_t = basis_t dL/drgb on the unclamped channels;
                      dL/ddir = sum_t grad basis_t (sh_t . dL/drgb);  dir = (mean - campos) / |.|: dL/dmean += (I - dir dir^T) / |.| dL/ddir
  cov3D -> s, q       Sigma = M M^T, M = R(q) diag(mod s):  dL/dM = 2 Gs M (Gs: symmetric, off-diagonals = half the both-entries
                      derivative);  dL/ds_j = mod sum_i R_ij dL/dM_ij;  dL/dR_ij = mod s_j dL/dM_ij;  R(q) of an UN-normalised
                      quaternion (w, x, y, z) as the rasterizer receives it (utils/general_utils.py:137-140 normalises before).

One published detail is NOT reproduced, here or in the kernels: upstream divides by det^2 + 1e-7 in the first step.  det >= 0.09
(the 0.3 dilation), so that changes dL/dcov2D by at most 1.2e-5 relative -- below the 1e-4 bar; the oracle and the kernels
use the exact derivative.

All arrays float64, vectorised over Gaussians.  ``tests/test_oracle_cpu.py::test_autograd_preprocess_backward_equals_the_hand_derivation``
holds the oracle's means3D / scales / rotations / shs / opacities gradients and its four camera gradients against this, in both
clamp semantics, on a scene in which about a hundred frustum-clamped Gaussians reach the image."""
import numpy as np

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis_and_gradient(d, deg):
    """basis (n, M) and its gradient w.r.t. the (unit) direction (n, M, 3); sign conventions of utils/sh_utils.py:57-112."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    n = d.shape[0]
    M = (deg + 1) ** 2
    B = np.zeros((n, M)); G = np.zeros((n, M, 3))
    B[:, 0] = C0
    if deg >= 1:
        B[:, 1] = -C1 * y; G[:, 1, 1] = -C1
        B[:, 2] = C1 * z;  G[:, 2, 2] = C1
        B[:, 3] = -C1 * x; G[:, 3, 0] = -C1
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        B[:, 4] = C2[0] * x * y;                 G[:, 4, 0] = C2[0] * y;  G[:, 4, 1] = C2[0] * x
        B[:, 5] = C2[1] * y * z;                 G[:, 5, 1] = C2[1] * z;  G[:, 5, 2] = C2[1] * y
        B[:, 6] = C2[2] * (2 * zz - xx - yy);    G[:, 6, 0] = -2 * C2[2] * x; G[:, 6, 1] = -2 * C2[2] * y; G[:, 6, 2] = 4 * C2[2] * z
        B[:, 7] = C2[3] * x * z;                 G[:, 7, 0] = C2[3] * z;  G[:, 7, 2] = C2[3] * x
        B[:, 8] = C2[4] * (xx - yy);             G[:, 8, 0] = 2 * C2[4] * x;  G[:, 8, 1] = -2 * C2[4] * y
    if deg >= 3:
        B[:, 9] = C3[0] * y * (3 * xx - yy)
        G[:, 9, 0] = C3[0] * 6 * x * y;  G[:, 9, 1] = C3[0] * (3 * xx - 3 * yy)
        B[:, 10] = C3[1] * x * y * z
        G[:, 10, 0] = C3[1] * y * z;  G[:, 10, 1] = C3[1] * x * z;  G[:, 10, 2] = C3[1] * x * y
        B[:, 11] = C3[2] * y * (4 * zz - xx - yy)
        G[:, 11, 0] = C3[2] * (-2 * x * y);  G[:, 11, 1] = C3[2] * (4 * zz - xx - 3 * yy);  G[:, 11, 2] = C3[2] * 8 * y * z
        B[:, 12] = C3[3] * z * (2 * zz - 3 * xx - 3 * yy)
        G[:, 12, 0] = C3[3] * (-6 * x * z);  G[:, 12, 1] = C3[3] * (-6 * y * z);  G[:, 12, 2] = C3[3] * (6 * zz - 3 * xx - 3 * yy)
        B[:, 13] = C3[4] * x * (4 * zz - xx - yy)
        G[:, 13, 0] = C3[4] * (4 * zz - 3 * xx - yy);  G[:, 13, 1] = C3[4] * (-2 * x * y);  G[:, 13, 2] = C3[4] * 8 * x * z
        B[:, 14] = C3[5] * z * (xx - yy)
        G[:, 14, 0] = C3[5] * 2 * x * z;  G[:, 14, 1] = C3[5] * (-2 * y * z);  G[:, 14, 2] = C3[5] * (xx - yy)
        B[:, 15] = C3[6] * x * (xx - 3 * yy)
        G[:, 15, 0] = C3[6] * (3 * xx - 3 * yy);  G[:, 15, 1] = C3[6] * (-6 * x * y)
    return B, G


def rotation_of(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z);     R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z);     R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y);     R[:, 2, 1] = 2 * (y * z + r * x);     R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def preprocess_backward(means3D, scales, rotations, shs, viewmatrix, projmatrix, intrinsic, campos, W, H, tanfovx, tanfovy,
                        scale_modifier, sh_degree, live, g_xy, g_conic, g_opacity, g_rgb, clamp_grad="stock", det_reg=0.0):
    """``live`` (P,) bool: the Gaussians that took part (raster_oracle: near-plane survivors); g_* the 2-D gradients the
    blend backward accumulated (pixel units; g_conic[:, 1] w.r.t. the conic's b itself).  Returns the gradients of means3D,
    scales, rotations, shs, opacities, of the four camera tensors (viewmatrix, projmatrix, intrinsic, campos; shift_factors = 0),
    and the mask of frustum-clamped Gaussians.  ``det_reg``: upstream's 1e-7 added to det^2 in the conic step (decision D9: 0 here
    and in the kernels; the test measures what the difference amounts to)."""
    P = means3D.shape[0]
    v, m, k = viewmatrix.reshape(4, 4), projmatrix.reshape(4, 4), intrinsic.reshape(4, 4)
    d_mean = np.zeros((P, 3)); d_scale = np.zeros((P, 3)); d_rot = np.zeros((P, 4)); d_sh = np.zeros_like(shs)
    idx = np.nonzero(live)[0]
    p = means3D[idx]
    gxy, gcon, grgb = g_xy[idx], g_conic[idx], g_rgb[idx]
    Rv = v[:3, :3]                                          # R[k][c]: t_c = sum_k p_k R[k][c] + v[3][c]
    t = p @ Rv + v[3, :3]
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    fx, fy = k[0, 0] * 0.5 * W, k[1, 1] * 0.5 * H           # D1
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    clx, cly = np.abs(tx / tz) > limx, np.abs(ty / tz) > limy
    txc = np.clip(tx / tz, -limx, limx) * tz
    tyc = np.clip(ty / tz, -limy, limy) * tz
    # ---- forward pieces needed again
    mod = float(scale_modifier)
    Rq = rotation_of(rotations[idx])
    Mm = Rq * (scales[idx] * mod)[:, None, :]               # M_ij = R_ij s_j
    Sig = Mm @ Mm.transpose(0, 2, 1)
    J00, J02, J11, J12 = fx / tz, -fx * txc / tz ** 2, fy / tz, -fy * tyc / tz ** 2
    T0 = J00[:, None] * Rv[:, 0][None, :] + J02[:, None] * Rv[:, 2][None, :]          # T0k = J00 R[k][0] + J02 R[k][2]
    T1 = J11[:, None] * Rv[:, 1][None, :] + J12[:, None] * Rv[:, 2][None, :]
    a = np.einsum("ni,nij,nj->n", T0, Sig, T0) + 0.3
    b = np.einsum("ni,nij,nj->n", T0, Sig, T1)
    c = np.einsum("ni,nij,nj->n", T1, Sig, T1) + 0.3
    det = a * c - b * b
    # ---- conic -> cov2D
    gA, gB, gC = gcon[:, 0], gcon[:, 1], gcon[:, 2]
    inv2 = 1.0 / (det * det + det_reg)
    dLa = inv2 * (-c * c * gA + b * c * gB - b * b * gC)
    dLc = inv2 * (-a * a * gC + a * b * gB - b * b * gA)
    dLb = inv2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
    # ---- cov2D -> Sigma (symmetric matrix of derivatives; off-diagonal entries hold half of the both-entries derivative)
    Gs = (dLa[:, None, None] * T0[:, :, None] * T0[:, None, :] + dLc[:, None, None] * T1[:, :, None] * T1[:, None, :]
          + 0.5 * dLb[:, None, None] * (T0[:, :, None] * T1[:, None, :] + T1[:, :, None] * T0[:, None, :]))
    # ---- cov2D -> T -> J -> t -> mean
    ST0, ST1 = np.einsum("nij,nj->ni", Sig, T0), np.einsum("nij,nj->ni", Sig, T1)
    dT0 = 2 * ST0 * dLa[:, None] + ST1 * dLb[:, None]
    dT1 = 2 * ST1 * dLc[:, None] + ST0 * dLb[:, None]
    dJ00, dJ02 = dT0 @ Rv[:, 0], dT0 @ Rv[:, 2]
    dJ11, dJ12 = dT1 @ Rv[:, 1], dT1 @ Rv[:, 2]
    itz2, itz3 = 1.0 / tz ** 2, 1.0 / tz ** 3
    if clamp_grad == "stock":
        dtx = np.where(clx, 0.0, 1.0) * (-fx * itz2) * dJ02
        dty = np.where(cly, 0.0, 1.0) * (-fy * itz2) * dJ12
        dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + 2 * fx * txc * itz3 * dJ02 + 2 * fy * tyc * itz3 * dJ12
    else:                                                   # tx' = +-lim tz where clamped: J02 = -+fx lim / tz
        dtx = np.where(clx, 0.0, 1.0) * (-fx * itz2) * dJ02
        dty = np.where(cly, 0.0, 1.0) * (-fy * itz2) * dJ12
        dtz = (-fx * itz2 * dJ00 - fy * itz2 * dJ11 + np.where(clx, 1.0, 2.0) * fx * txc * itz3 * dJ02
               + np.where(cly, 1.0, 2.0) * fy * tyc * itz3 * dJ12)
    dm = dtx[:, None] * Rv[:, 0][None, :] + dty[:, None] * Rv[:, 1][None, :] + dtz[:, None] * Rv[:, 2][None, :]
    # ---- pixel -> mean
    hom = p @ m[:3, :] + m[3, :]
    mw = 1.0 / (hom[:, 3] + 1e-7)
    gnx, gny = gxy[:, 0] * (0.5 * W), gxy[:, 1] * (0.5 * H)
    mul1, mul2 = hom[:, 0] * mw * mw, hom[:, 1] * mw * mw
    for kk in range(3):
        dm[:, kk] += (m[kk, 0] * mw - m[kk, 3] * mul1) * gnx + (m[kk, 1] * mw - m[kk, 3] * mul2) * gny
    # ---- colour
    if shs is not None:
        dv = p - campos[None, :]
        ln = np.sqrt((dv * dv).sum(1))
        dirn = dv / ln[:, None]
        B, Gd = sh_basis_and_gradient(dirn, sh_degree)
        M = B.shape[1]
        sh = shs[idx][:, :M, :]
        raw = np.einsum("nt,ntc->nc", B, sh) + 0.5
        gr = np.where(raw < 0, 0.0, grgb)                   # clamped channels pass nothing
        d_sh_l = np.zeros_like(shs[idx])
        d_sh_l[:, :M, :] = B[:, :, None] * gr[:, None, :]
        d_sh[idx] = d_sh_l
        ddir = np.einsum("ntk,ntc,nc->nk", Gd, sh, gr)
        dm_dir = (ddir - dirn * (dirn * ddir).sum(1)[:, None]) / ln[:, None]
        dm += dm_dir
    d_mean[idx] = dm
    # ---- the fork's camera gradients (no published counterpart: the same chain rule continued into the matrices)
    p1 = np.concatenate([p, np.ones((idx.size, 1))], 1)                               # [x y z 1]
    d_view = np.zeros((4, 4)); d_proj = np.zeros((4, 4)); d_intr = np.zeros((4, 4))
    d_view[:, 0] = p1.T @ dtx; d_view[:, 1] = p1.T @ dty; d_view[:, 2] = p1.T @ dtz    # through t = [p 1] . viewmatrix
    d_view[:3, 0] += (J00[:, None] * dT0).sum(0)                                      # through T = J W
    d_view[:3, 1] += (J11[:, None] * dT1).sum(0)
    d_view[:3, 2] += (J02[:, None] * dT0 + J12[:, None] * dT1).sum(0)
    d_proj[:, 0] = p1.T @ (mw * gnx); d_proj[:, 1] = p1.T @ (mw * gny); d_proj[:, 3] = p1.T @ (-(mul1 * gnx + mul2 * gny))
    d_intr[0, 0] = 0.5 * W * (dJ00 / tz - dJ02 * txc * itz2).sum()                    # fx = intrinsic[0][0] W / 2 (D1)
    d_intr[1, 1] = 0.5 * H * (dJ11 / tz - dJ12 * tyc * itz2).sum()
    d_campos = -(dm_dir.sum(0)) if shs is not None else np.zeros(3)
    # ---- Sigma -> scale, quaternion
    dM = 2 * Gs @ Mm
    s_mod = scales[idx] * mod
    d_scale[idx] = mod * (Rq * dM).sum(1)
    g = dM * s_mod[:, None, :]                              # dL/dR_ij
    r, x, y, z = [rotations[idx][:, i] for i in range(4)]
    dq = np.empty((idx.size, 4))
    dq[:, 0] = 2 * (-z * g[:, 0, 1] + y * g[:, 0, 2] + z * g[:, 1, 0] - x * g[:, 1, 2] - y * g[:, 2, 0] + x * g[:, 2, 1])
    dq[:, 1] = 2 * (y * g[:, 0, 1] + z * g[:, 0, 2] + y * g[:, 1, 0] - 2 * x * g[:, 1, 1] - r * g[:, 1, 2] + z * g[:, 2, 0]
                    + r * g[:, 2, 1] - 2 * x * g[:, 2, 2])
    dq[:, 2] = 2 * (-2 * y * g[:, 0, 0] + x * g[:, 0, 1] + r * g[:, 0, 2] + x * g[:, 1, 0] + z * g[:, 1, 2] - r * g[:, 2, 0]
                    + z * g[:, 2, 1] - 2 * y * g[:, 2, 2])
    dq[:, 3] = 2 * (-2 * z * g[:, 0, 0] - r * g[:, 0, 1] + x * g[:, 0, 2] + r * g[:, 1, 0] - 2 * z * g[:, 1, 1] + y * g[:, 1, 2]
                    + x * g[:, 2, 0] + y * g[:, 2, 1])
    d_rot[idx] = dq
    clamped = np.zeros(P, dtype=bool); clamped[idx] = clx | cly
    # ---- the forward values of the same Gaussians (A.1), for the comparison with raster_oracle.preprocess
    fwd = dict(idx=idx, conic=np.stack([c / det, -b / det, a / det], 1),
               xy=np.stack([((hom[:, 0] * mw + 1.0) * W - 1.0) * 0.5, ((hom[:, 1] * mw + 1.0) * H - 1.0) * 0.5], 1),
               radius=np.ceil(3.0 * np.sqrt(0.5 * (a + c) + np.sqrt(np.maximum(0.1, (0.5 * (a + c)) ** 2 - det)))),
               rgb=np.maximum(raw, 0.0) if shs is not None else None, depth=tz)
    return dict(means3D=d_mean, scales=d_scale, rotations=d_rot, shs=d_sh, opacities=g_opacity.reshape(P, 1).copy(), clamped=clamped,
                viewmatrix=d_view, projmatrix=d_proj, intrinsic=d_intr, campos=d_campos, forward=fwd)
