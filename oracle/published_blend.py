"""Per-pixel loops of the published 3DGS tile blend (forward AND the hand-derived backward recurrences)  --  TEST
INFRASTRUCTURE ONLY (same rule as raster_oracle.py: tests import it, the product never does).

Why a second restatement.  ``raster_oracle.py`` gets every gradient from autograd on dense (pixel x splat) tensors: nothing
in it is hand-derived, which makes it an independent check of the HIP backward -- but also means that the oracle's backward
has never been held against the recurrences the published rasterizer actually executes.  The reference's own rasterizer
(``diff_gaussian_rasterization`` = denghilbert/3dgs-pose @ cd77ced, /root/reference/README.md:126) is an empty submodule, so
those recurrences cannot be read from /root/reference; they are restated here from the published algorithm (SURVEY.md
Appendix A.3 / A.4, [UPSTREAM-KNOWLEDGE]) as plain sequential loops in float64, one pixel and one list entry at a time, in the
order the published kernels walk them:

  forward (A.3)   T = 1; for each entry front to back: d = centre - pixel; power = -1/2 (a dx^2 + c dy^2) - b dx dy;
                  power > 0: skip; alpha = min(0.99, o exp(power)); alpha < 1/255: skip; T' = T (1 - alpha); T' < 1e-4: stop
                  BEFORE blending this entry; C += rgb alpha T; T = T'; n_contrib = position of the last blended entry.
  backward (A.4)  T = T_final; walk back from n_contrib with the same skips; T = T / (1 - alpha);
                  dL/drgb += alpha T dL/dC;   accum = last_alpha last_rgb + (1 - last_alpha) accum  (colour seen behind);
                  dL/dalpha = T sum_c (rgb_c - accum_c) dL/dC_c  -  T_final / (1 - alpha) sum_c bg_c dL/dC_c;
                  dL/dG = o dL/dalpha (straight through the 0.99 clamp, decision D3);  dL/do += G dL/dalpha;
                  dL/dcentre += dL/dG G (-(a dx + b dy), -(c dy + b dx));
                  dL/d(a, b, c) += dL/dG G (-1/2 dx^2, -dx dy, -1/2 dy^2).
                  (The published kernel keeps HALF of the b term in its intermediate and doubles it in the covariance
                  backward; what is compared here is the derivative itself.)
                  fork addition (D4): densify += |dL/dcentre of this pixel| (W/2, H/2), summed over pixels.

``tests/test_oracle_cpu.py::test_autograd_blend_equals_the_published_recurrences`` feeds both restatements the same
preprocessed 2-D splats and lists and requires image, n_contrib, final_T and all five gradient sets to agree to float64
round-off.  Sizes: a few thousand (pixel, entry) pairs per tile -- pure Python, seconds."""
import numpy as np

TILE = 16
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4


def blend_forward(xy, conic, opacity, rgb, point_list, ranges, bg, W, H):
    """A.3.  xy (P,2) pixel centres, conic (P,3) = (a, b, c), opacity (P,), rgb (P,3), ranges (T,2) into point_list.
    Returns image (3,H,W), final_T (H,W), n_contrib (H,W) (1-based position of the last blended entry, 0 = none)."""
    gx = (W + TILE - 1) // TILE
    image = np.zeros((3, H, W)); final_T = np.ones((H, W)); n_contrib = np.zeros((H, W), dtype=np.int64)
    for py in range(H):
        for px in range(W):
            t = (py // TILE) * gx + px // TILE
            lo, hi = int(ranges[t, 0]), int(ranges[t, 1])
            T = 1.0; C = np.zeros(3); last = 0
            for k in range(lo, hi):
                g = int(point_list[k])
                dx, dy = xy[g, 0] - px, xy[g, 1] - py
                a, b, c = conic[g]
                power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
                if power > 0.0:
                    continue
                alpha = min(0.99, opacity[g] * np.exp(power))
                if alpha < ALPHA_MIN:
                    continue
                Tn = T * (1.0 - alpha)
                if Tn < T_STOP:
                    break
                C += rgb[g] * (alpha * T)
                T = Tn
                last = k - lo + 1
            image[:, py, px] = C + T * bg
            final_T[py, px] = T
            n_contrib[py, px] = last
    return image, final_T, n_contrib


def blend_backward(xy, conic, opacity, rgb, point_list, ranges, bg, W, H, final_T, n_contrib, grad_image):
    """A.4 with the recurrences as published.  Returns dict(xy (P,2), conic (P,3), opacity (P,), rgb (P,3), absgrad (P,2))."""
    P = xy.shape[0]
    gx = (W + TILE - 1) // TILE
    d_xy = np.zeros((P, 2)); d_conic = np.zeros((P, 3)); d_op = np.zeros(P); d_rgb = np.zeros((P, 3)); absgrad = np.zeros((P, 2))
    for py in range(H):
        for px in range(W):
            t = (py // TILE) * gx + px // TILE
            lo = int(ranges[t, 0])
            dL_dC = grad_image[:, py, px]
            Tf = final_T[py, px]
            bg_dot = float(np.dot(bg, dL_dC))
            T = Tf
            accum = np.zeros(3); last_alpha = 0.0; last_rgb = np.zeros(3)
            for k in range(lo + int(n_contrib[py, px]) - 1, lo - 1, -1):
                g = int(point_list[k])
                dx, dy = xy[g, 0] - px, xy[g, 1] - py
                a, b, c = conic[g]
                power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
                if power > 0.0:
                    continue
                G = np.exp(power)
                alpha = min(0.99, opacity[g] * G)
                if alpha < ALPHA_MIN:
                    continue
                T = T / (1.0 - alpha)
                d_rgb[g] += (alpha * T) * dL_dC
                accum = last_alpha * last_rgb + (1.0 - last_alpha) * accum
                last_rgb = rgb[g].copy()
                dL_dalpha = T * float(np.dot(rgb[g] - accum, dL_dC))
                last_alpha = alpha
                dL_dalpha += (-Tf / (1.0 - alpha)) * bg_dot
                dL_dG = opacity[g] * dL_dalpha
                gdx, gdy = G * dx, G * dy
                ex = dL_dG * (-gdx * a - gdy * b)
                ey = dL_dG * (-gdy * c - gdx * b)
                d_xy[g, 0] += ex; d_xy[g, 1] += ey
                absgrad[g, 0] += abs(ex * (0.5 * W)); absgrad[g, 1] += abs(ey * (0.5 * H))
                d_conic[g, 0] += -0.5 * gdx * dx * dL_dG
                d_conic[g, 1] += -1.0 * gdx * dy * dL_dG
                d_conic[g, 2] += -0.5 * gdy * dy * dL_dG
                d_op[g] += G * dL_dalpha
    return dict(xy=d_xy, conic=d_conic, opacity=d_op, rgb=d_rgb, absgrad=absgrad)
