"""D7, second half (tile masks): a tile dropped from a Gaussian's rectangle must hold no pixel with alpha >= 1/255.

CPU property test on the oracle's masks (which the HIP kernel reproduces bit for bit, tests/test_parity_gpu.py): for
scenes of ordinary, needle-shaped and nearly opaque splats, every pixel centre of every dropped tile is evaluated in
float64 from the conic / centre / opacity the blend kernels use.  The margin of the rule (2 % + 0.022 on 2 ln(255 o)) must
leave the largest such alpha clearly below the threshold."""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as O
from scenes import make_case, oracle_settings


def _dropped_tile_alpha(scene, cam, deg, P):
    s = oracle_settings(cam, deg)
    with torch.no_grad():
        pre = O.preprocess(scene["means3D"], torch.zeros(P, 3), torch.zeros(3), scene["shs"], None, scene["opacities"],
                           scene["scales"], scene["rotations"], None, s, torch.float32, None)
    r = pre.rect.long()
    w, h = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
    small = (pre.radii > 0) & (w > 0) & (h > 0) & (w <= 8) & (h <= 8)
    idx = torch.nonzero(small).reshape(-1)
    xy, con, op = pre.xy.double(), pre.conic.double(), pre.opacity.double()
    jj, ii = np.meshgrid(np.arange(16.0), np.arange(16.0), indexing="xy")
    jj, ii = torch.tensor(jj.reshape(-1)), torch.tensor(ii.reshape(-1))
    worst, ndrop, nrect = 0.0, 0, int((w * h)[idx].sum())
    for ry in range(8):
        for rx in range(8):
            live = (rx < w[idx]) & (ry < h[idx]) & (((pre.keep[idx] >> (ry * 8 + rx)) & 1) == 0)
            g = idx[live]
            if g.numel() == 0:
                continue
            X0, Y0 = (r[g, 0] + rx).double() * 16, (r[g, 1] + ry).double() * 16
            dx = X0[:, None] + jj[None] - xy[g, 0:1]
            dy = Y0[:, None] + ii[None] - xy[g, 1:2]
            q = con[g, 0:1] * dx * dx + 2 * con[g, 1:2] * dx * dy + con[g, 2:3] * dy * dy
            alpha = torch.minimum(torch.tensor(0.99, dtype=torch.float64), op[g][:, None] * torch.exp(-0.5 * q))
            alpha = torch.where(q >= 0, alpha, torch.zeros_like(alpha))          # power > 0 is skipped by the blend
            worst = max(worst, float(alpha.max()))
            ndrop += int(g.numel())
    # the masks and tiles_touched must agree
    pop = torch.zeros_like(pre.keep)
    for b in range(64):
        pop += (pre.keep >> b) & 1
    assert torch.equal(pop[small], pre.tiles_touched.long()[small])
    return worst, ndrop, nrect


@pytest.mark.parametrize("kind", ["ordinary", "needles", "opaque_large", "faint"])
def test_dropped_tiles_hold_no_contributing_pixel(kind):
    P, W, H = 4000, 512, 384
    scene, cam = make_case(P, W, H, 1.0, 1, seed=123)
    g = torch.Generator().manual_seed(7)
    if kind == "needles":
        length = torch.exp(torch.empty(P, 1).uniform_(-2.8, -1.2, generator=g))
        scene["scales"] = torch.cat([length, length / 50.0, length / 50.0], 1)[:, torch.randperm(3, generator=g)]
        q = torch.randn(P, 4, generator=g)
        scene["rotations"] = q / q.norm(dim=1, keepdim=True)
    elif kind == "opaque_large":
        scene["scales"] = scene["scales"] * 3.0
        scene["opacities"] = torch.full((P, 1), 0.999)
    elif kind == "faint":
        scene["opacities"] = (1.0 / 255.0) * (1.0 + 3.0 * torch.rand(P, 1, generator=g))      # barely above the threshold
    worst, ndrop, nrect = _dropped_tile_alpha(scene, cam, 1, P)
    print(kind, "dropped", ndrop, "of", nrect, "tiles of small rectangles; max alpha in a dropped tile", worst)
    assert ndrop > 0.05 * nrect, (ndrop, nrect)
    assert worst < (1.0 / 255.0) * 0.995, worst
