"""CPU analysis (not a test): how many (tile, Gaussian) instances of the bench scene can never contribute, and a
per-tile-row interval rule that finds them with (rows + 1) square roots per Gaussian.

A splat contributes to a pixel only where alpha = o exp(-d^T Sigma^-1 d / 2) >= 1/255, i.e. inside the ellipse
d^T Sigma^-1 d <= T2 = 2 ln(255 o).  With Sigma = [[cxx, cxy], [cxy, cyy]] (no inverse needed):

    x-extent of the ellipse at vertical offset dy:   dx = (cxy dy +- sqrt(det (T2 cyy - dy^2))) / cyy     (dy^2 <= T2 cyy)
    leftmost / rightmost point:                      dx = -+ sqrt(T2 cxx)  at  dy = cxy / cxx * dx

The left boundary is convex in dy and the right one concave, so over the band of a tile row (pixel centres 16 ty ... 16 ty
+ 15) the extent is attained at the band's two ends or at the extreme point if it lies inside the band.  The rule keeps
tile (tx, ty) of the Gaussian's rectangle iff [16 tx, 16 tx + 15] meets that extent (plus a margin).

The script checks, on the bench scene (config 3):
  * the rule never drops an instance the exact ellipse / pixel-square test keeps (conservative),
  * nor one with a pixel of alpha >= 1/255 by brute force over the 256 pixel centres (sampled instances),
  * and prints the share of instances it drops next to the exact test's and today's 16-block reach masks'.

usage: python tests/analysis_reach_rows.py [P] [W] [H] [sm]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from analysis_lane_fill import reach_masks  # noqa: E402
from oracle import raster_oracle as O  # noqa: E402
from scenes import make_case, oracle_settings  # noqa: E402

REL, ABS = 1.02, 0.1        # the margins preprocess_fwd uses for the opacity-aware rectangle: 2 % on T2, 0.1 px


def row_rule(px, py, cxx, cxy, cyy, T2, tx, ty):
    """Keep mask of the per-row interval rule for instances (tile tx, ty) of Gaussians with the given 2-D state."""
    det = cxx * cyy - cxy * cxy
    T2m = T2 * REL
    hx, hy = np.sqrt(T2m * cxx), np.sqrt(T2m * cyy)
    dyA, dyB = 16.0 * ty - py, 16.0 * ty + 15.0 - py

    def ends(dy):
        inside = dy * dy <= T2m * cyy
        root = np.sqrt(np.maximum(det * (T2m * cyy - dy * dy), 0.0))
        lo = np.where(inside, (cxy * dy - root) / cyy, np.inf)
        hi = np.where(inside, (cxy * dy + root) / cyy, -np.inf)
        return lo, hi
    loA, hiA = ends(dyA)
    loB, hiB = ends(dyB)
    lo, hi = np.minimum(loA, loB), np.maximum(hiA, hiB)
    dyL, dyR = cxy / cxx * -hx, cxy / cxx * hx              # vertical offsets of the leftmost / rightmost point
    lo = np.where((dyL >= dyA) & (dyL <= dyB), -hx, lo)
    hi = np.where((dyR >= dyA) & (dyR <= dyB), hx, hi)
    # a band that contains the whole ellipse vertically has both extreme points inside it, so lo / hi are set above;
    # a band that misses it on one side keeps lo = +inf, hi = -inf: nothing kept
    x0, x1 = 16.0 * tx - px, 16.0 * tx + 15.0 - px
    return (x1 >= lo - ABS) & (x0 <= hi + ABS)


def exact_reach(px, py, a, b, c, T2, X0, Y0):
    """min of the conic quadratic form over the square of pixel centres [X0, X0+15] x [Y0, Y0+15] <= T2 (float64)."""
    X1, Y1 = X0 + 15.0, Y0 + 15.0

    def edge_x(yv):
        dy = yv - py
        dx = np.clip(px - b * dy / a, X0, X1) - px
        return a * dx * dx + 2 * b * dx * dy + c * dy * dy

    def edge_y(xv):
        dx = xv - px
        dy = np.clip(py - b * dx / c, Y0, Y1) - py
        return a * dx * dx + 2 * b * dx * dy + c * dy * dy
    inside = (px >= X0) & (px <= X1) & (py >= Y0) & (py <= Y1)
    qm = np.minimum(np.minimum(edge_x(Y0), edge_x(Y1)), np.minimum(edge_y(X0), edge_y(X1)))
    return inside | (qm <= T2)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    sm = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
    scene, cam = make_case(P, W, H, sm, 3, seed=0)
    s = oracle_settings(cam, 3)
    with torch.no_grad():
        pre = O.preprocess(scene["means3D"], torch.zeros(P, 3), torch.zeros(3), scene["shs"], None, scene["opacities"],
                           scene["scales"], scene["rotations"], None, s, torch.float32, None)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    _, pl, ranges, _ = O.bin_and_sort(pre.depth.float(), pre.rect, pre.tiles_touched, gx, gy, pre.keep)
    pl = pl.numpy().astype(np.int64)
    ranges = ranges.numpy().astype(np.int64)
    tile = np.repeat(np.arange(gx * gy), ranges[:, 1] - ranges[:, 0])
    tx, ty = (tile % gx).astype(np.float64), (tile // gx).astype(np.float64)
    xy, conic, op = pre.xy.numpy().astype(np.float64), pre.conic.numpy().astype(np.float64), pre.opacity.numpy().astype(np.float64)
    px, py = xy[pl, 0], xy[pl, 1]
    a, b, c = conic[pl, 0], conic[pl, 1], conic[pl, 2]
    dc = a * c - b * b
    cxx, cxy, cyy = c / dc, -b / dc, a / dc                         # cov2D from its inverse
    T2 = 2.0 * np.log(np.maximum(255.0 * op[pl], 1.0))
    keep_rows = row_rule(px, py, cxx, cxy, cyy, T2, tx, ty)
    keep_exact = exact_reach(px, py, a, b, c, T2, 16.0 * tx, 16.0 * ty)
    keep_blocks = reach_masks(px, py, a, b, c, op[pl], 16.0 * tx, 16.0 * ty).any(1)
    n = pl.size
    print(f"P={P} {W}x{H} sm={sm}: I = {n} (opacity-aware rectangles)")
    print(f"  dropped by the exact ellipse / pixel-square test : {1 - keep_exact.mean():.4f}")
    print(f"  dropped by the per-row interval rule            : {1 - keep_rows.mean():.4f}")
    print(f"  dropped by today's 16-block reach masks (blend)  : {1 - keep_blocks.mean():.4f}")
    bad = keep_exact & ~keep_rows
    print(f"  instances the rule drops but the exact test keeps: {int(bad.sum())}   (must be 0)")
    # brute force over the 256 pixel centres on a sample of the dropped instances
    rng = np.random.default_rng(0)
    idx = np.flatnonzero(~keep_rows)
    idx = rng.choice(idx, size=min(20000, idx.size), replace=False)
    ii, jj = np.meshgrid(np.arange(16.0), np.arange(16.0), indexing="ij")
    dx = (16.0 * tx[idx])[:, None] + jj.reshape(1, -1) - px[idx][:, None]
    dy = (16.0 * ty[idx])[:, None] + ii.reshape(1, -1) - py[idx][:, None]
    q = a[idx][:, None] * dx * dx + 2 * b[idx][:, None] * dx * dy + c[idx][:, None] * dy * dy
    alpha = np.minimum(0.99, op[pl][idx][:, None] * np.exp(-0.5 * q))
    worst = alpha.max()
    print(f"  brute force on {idx.size} dropped instances: max alpha over their tiles' pixels = {worst:.6f} "
          f"(threshold 1/255 = {1 / 255:.6f}) -> {'ok' if worst < 1 / 255 else 'VIOLATION'}")
    return 0 if (bad.sum() == 0 and worst < 1 / 255) else 1


if __name__ == "__main__":
    sys.exit(main())
