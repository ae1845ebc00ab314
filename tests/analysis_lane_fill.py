"""CPU analysis (not a test): lane fill of the backward blend's group phase on the bench scene.

The backward (csrc/blend.hip, blend_bwd_scan_kernel) gives every 16-lane DPP row of a wave one 4x4 block of the tile and
walks that block's list 16 splats per step; a wave runs max over its four rows of ceil(L_block / 16) steps per chunk.
This script rebuilds the per-(tile, chunk, block) list lengths from the oracle's sorted lists and the kernel's reach-mask
rule and prints how many of the issued lane slots hold a live (block, splat) pair under a few assignment policies.

usage: python tests/analysis_lane_fill.py [--ring | --fwd] [P] [W] [H] [sm]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import raster_oracle as O  # noqa: E402
from scenes import make_case, oracle_settings  # noqa: E402


def reach_masks(x, y, a, b, c, o, X0, Y0):
    """numpy restatement of block_mask16 (csrc/blend.hip): (n,16) bool."""
    vis = 255.0 * o
    det = a * c - b * b
    tau2 = 2.0 * (np.maximum(np.log(np.maximum(vis, 1e-30)), 0.0) + 1e-3)
    hx = np.sqrt(tau2 * c / det) * 1.001 + 0.05
    hy = np.sqrt(tau2 * a / det) * 1.001 + 0.05
    qd, qo = 2.25 * (a + c), 4.5 * b
    rb = np.sqrt(np.maximum(qd + qo, qd - qo))
    lim2 = (np.sqrt(tau2) * 1.001 + rb + 1e-3) ** 2
    xl, xh, yl, yh = x - hx - X0, x + hx - X0, y - hy - Y0, y + hy - Y0
    ex, ey = (X0 + 1.5) - x, (Y0 + 1.5) - y
    m = np.zeros((x.size, 16), dtype=bool)
    for by in range(4):
        dy = ey + 4.0 * by
        rowok = (yh >= 4.0 * by) & (yl <= 4.0 * by + 3.0)
        for bx in range(4):
            dx = ex + 4.0 * bx
            Q = dx * (a * dx + 2.0 * b * dy) + c * dy * dy
            m[:, by * 4 + bx] = rowok & (xh >= 4.0 * bx) & (xl <= 4.0 * bx + 3.0) & (Q <= lim2)
    m[~((det > 0) & (a > 0) & (c > 0))] = True
    m[~(vis >= 0.99)] = False
    return m


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    sm = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
    scene, cam = make_case(P, W, H, sm, 3, seed=0)
    s = oracle_settings(cam, 3)
    with torch.no_grad():
        Pn = scene["means3D"].shape[0]
        pre = O.preprocess(scene["means3D"], torch.zeros(Pn, 3), torch.zeros(3), scene["shs"], None, scene["opacities"], scene["scales"],
                           scene["rotations"], None, s, torch.float32, None)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    _, pl, ranges, _ = O.bin_and_sort(pre.depth.float(), pre.rect, pre.tiles_touched, gx, gy, pre.keep)
    pl = pl.numpy().astype(np.int64)
    ranges = ranges.numpy().astype(np.int64)
    n_t = ranges[:, 1] - ranges[:, 0]
    I = pl.size
    tile = np.repeat(np.arange(gx * gy), n_t)
    pos = np.arange(I) - ranges[tile, 0]
    xy, conic, op = pre.xy.numpy().astype(np.float64), pre.conic.numpy().astype(np.float64), pre.opacity.numpy().astype(np.float64)
    X0, Y0 = (tile % gx) * 16.0, (tile // gx) * 16.0
    m = reach_masks(xy[pl, 0], xy[pl, 1], conic[pl, 0], conic[pl, 1], conic[pl, 2], op[pl], X0, Y0)
    print(f"P={P} {W}x{H} sm={sm}: I={I}, instances/tile mean {n_t.mean():.0f}, blocks reached per instance {m.sum(1).mean():.2f}")
    if globals().get("_RING"):
        ring_policies((m * (1 << np.arange(16))).sum(1).astype(np.uint16), ranges)
        return
    if globals().get("_FWD"):
        # forward (blend_fwd_rows_kernel): one splat per step and row, chunks of 255 cut from the FRONT, a wave runs the longest
        # of its four rows' lists per chunk: what the row imbalance costs and what other block -> wave groupings would give
        ck = pos // 255
        nck = int(ck.max()) + 1
        L = np.zeros((gx * gy, nck, 16), dtype=np.int64)
        for b in range(16):
            np.add.at(L[:, :, b], (tile[m[:, b]], ck[m[:, b]]), 1)
        live = L.sum()
        quad_idx = [[qy * 8 + qx * 2, qy * 8 + qx * 2 + 1, qy * 8 + qx * 2 + 4, qy * 8 + qx * 2 + 5] for qy in range(2) for qx in range(2)]
        quad = np.stack([L[:, :, q] for q in quad_idx], 2)
        st_quad = quad.max(3).sum()
        srt = np.sort(L, axis=2)[:, :, ::-1].reshape(gx * gy, nck, 4, 4)
        st_chunk = srt.max(3).sum()
        order = np.argsort(-L.sum(1), axis=1)                  # one grouping per tile, by the blocks' whole-tile list lengths
        Ls = np.take_along_axis(L, order[:, None, :].repeat(nck, 1), axis=2).reshape(gx * gy, nck, 4, 4)
        st_tile = Ls.max(3).sum()
        print(f"  forward: {live / 1e6:.2f} M (splat, block) entries = {live / 4e6:.2f} M wave steps at perfect balance; quadrant rows "
              f"{st_quad / 1e6:.2f} M ({live / 4 / st_quad:.3f}), ranked per tile {st_tile / 1e6:.2f} M ({live / 4 / st_tile:.3f}), "
              f"ranked per chunk {st_chunk / 1e6:.2f} M ({live / 4 / st_chunk:.3f})")
        return
    for chunk in (128, 256):
        ck = (n_t[tile] - 1 - pos) // chunk                    # chunks are cut from the back of the list
        nck = int(ck.max()) + 1
        L = np.zeros((gx * gy, nck, 16), dtype=np.int64)
        for b in range(16):
            np.add.at(L[:, :, b], (tile[m[:, b]], ck[m[:, b]]), 1)
        live = L.sum()
        for grp in (16, 8):
            steps = -(-L // grp)                              # per (tile, chunk, block)
            quant = (steps * grp).sum()
            # kernel today: wave = quadrant, rows = its four blocks, steps = max over rows
            quad = np.stack([steps[:, :, [qy * 8 + qx * 2, qy * 8 + qx * 2 + 1, qy * 8 + qx * 2 + 4, qy * 8 + qx * 2 + 5]]
                             for qy in range(2) for qx in range(2)], 2)            # (T, nck, wave, row)
            issued_quad = (quad.max(3) * 4 * grp).sum()
            # blocks ranked by list length, wave w takes ranks 4w..4w+3
            srt = np.sort(steps, axis=2)[:, :, ::-1].reshape(gx * gy, nck, 4, 4)
            issued_sorted = (srt.max(3) * 4 * grp).sum()
            wave_steps_quad = quad.max(3)                      # per-wave step counts -> barrier skew
            skew_quad = (wave_steps_quad.max(2) * 4).sum() / max(1, wave_steps_quad.sum())
            skew_sorted = (srt.max(3).max(2) * 4).sum() / max(1, srt.max(3).sum())
            print(f"  chunk {chunk:3d} group {grp:2d}: live {live / 1e6:.1f} M lane-steps; fill quantisation-only {live / quant:.3f}, "
                  f"quadrant rows {live / issued_quad:.3f} (slowest wave / mean {skew_quad:.2f}), "
                  f"length-ranked rows {live / issued_sorted:.3f} (slowest / mean {skew_sorted:.2f})")


def ring_policies(bits, ranges):
    """--ring: what carrying a block's partial 16-lane group over into the next chunk would buy (round 3, asked for by the
    round-2 review).  Ring of R staged slots; the last D of them stay staged into the next round ("tail zone"); a block
    walks its entries in the body and takes entries from the tail zone only to complete a group of 16.  D = 0 is the
    kernel's scheme (R = BCHUNK).  Prints wave steps (max over a wave's four rows per round), rounds and lane fill."""
    quad = [[qy * 8 + qx * 2, qy * 8 + qx * 2 + 1, qy * 8 + qx * 2 + 4, qy * 8 + qx * 2 + 5] for qy in range(2) for qx in range(2)]

    def sim(R, D):
        steps = rounds = live = 0
        for t in range(ranges.shape[0]):
            lo, hi = ranges[t]
            n = hi - lo
            if n == 0:
                continue
            mk = bits[lo:hi][::-1]                              # walk order: deepest first
            mb = ((mk[:, None] >> np.arange(16)) & 1).astype(bool)
            cum = np.concatenate([np.zeros((1, 16), np.int64), np.cumsum(mb, 0)])
            c = np.zeros(16, np.int64)                          # entries processed so far, per block
            F = 0
            while True:
                E = min(n, F + R)
                last = E == n
                body = E if last else E - D
                avail_body = np.maximum(cum[body] - c, 0)
                avail_all = cum[E] - c
                take = avail_all if last else np.where(cum[body] - c < 0, 0, avail_body + np.minimum((-avail_body) % 16, avail_all - avail_body))
                g = -(-take // 16)
                steps += sum(int(g[q].max()) for q in quad)
                live += int(take.sum())
                c = c + take
                rounds += 1
                if last:
                    break
                F = body
        return steps, rounds, live
    for R, D in ((128, 0), (128, 16), (128, 32), (128, 48), (128, 64), (64, 0), (96, 0), (160, 0), (176, 0), (256, 0), (10 ** 9, 0)):
        st, r, l = sim(R, D)
        print(f"  ring R={R if R < 10 ** 9 else 'whole tile'} D={D}: wave steps {st}, rounds {r}, lane fill {l / (st * 64):.3f}")


if __name__ == "__main__":
    if "--ring" in sys.argv:
        sys.argv.remove("--ring")
        _RING = True
    else:
        _RING = False
    _FWD = "--fwd" in sys.argv
    if _FWD:
        sys.argv.remove("--fwd")
    main()
