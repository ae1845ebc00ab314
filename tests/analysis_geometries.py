"""CPU analysis (not a test): other distributions of the backward blend's work, priced before anything is built.

Round-4 review, item 1: "two splats per lane" and "finer reach granularity (4x2-pixel blocks)" for blend_bwd_scan_kernel
(csrc/blend.hip).  This script rebuilds, from the oracle's sorted lists of the bench scene, what each geometry would
evaluate: (splat, sub-block) list entries, evaluated and contributing (pixel, splat) pairs, wave steps per chunk with the
quantisation of the lists into lane groups and the spread between the rows a wave runs in lockstep -- and prices a wave
step with the issue costs measured by tools/ubench/issue_rates.hip at three waves per SIMD (ns per wave instruction on gfx950:
plain 1.39, packed 1.96, DPP 1.93, transcendental 3.47).  Row step of the shipped kernel (4 pixels x 64 lanes, ISA count): 36 DPP,
~50 packed, 8 transcendental, ~40 plain = ~250 ns; per 16-splat step ~85 plain instructions of list / accumulator bookkeeping
(four read-modify-write phases into the wave's LDS copy); per chunk and wave ~170 + ~12 per (list, 64 staged slots).

Geometries (all: lane = splat, DPP affine scan over a lane group, 11 sums per lane):
  A   shipped: 4x4 blocks, 16-lane groups (4 scan steps), wave = quadrant = 4 rows, 4 row steps per entry
  E   two splats per lane: 4x4 blocks, 32 entries per row group, composed in the lane before the 16-lane scan
  B   4x4 blocks, 8-lane groups INTERLEAVED in a DPP row (even / odd lanes, row_shr:2/4/8 = 3 scan steps), wave = 8 blocks
  C   4x2 sub-blocks, 8-lane interleaved groups, wave = the 8 sub-blocks of a quadrant, 2 row steps per entry
  D   4x2 sub-blocks, 16-lane groups, wave = 4 sub-blocks, 2 row steps per entry (8 waves of work per tile)
  F   2x2 sub-blocks (64 lists per tile), 16-lane groups: one row step of 4 pixels per entry

usage: python tests/analysis_geometries.py [P] [W] [H] [sm]
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

# ns per wave instruction and SIMD at three waves per SIMD (profiles/r03/ubench_issue_rates.txt, tools/ubench/issue_rates.hip)
C_PLAIN, C_PK, C_DPP, C_TRANS = 1.39, 1.96, 1.93, 3.47
ROW_STEP = 36 * C_DPP + 50 * C_PK + 8 * C_TRANS + 40 * C_PLAIN          # ~250 ns, shipped row step (tools/ubench/mfma_outer.hip measures 250)
SCAN_STEP = 8 * C_DPP                                                    # one shift distance of the scan: 4 fmac + 4 mul
STEP_OVERHEAD = 85 * C_PLAIN                                             # fold + four RMW phases + list entry, per 16-splat step
PHASE = 11 * C_PLAIN                                                     # one more read-modify-write phase
CHUNK_OVERHEAD = 170 * C_PLAIN
LIST_ROUND = 12 * C_PLAIN                                                # ballot + popcount + store, per list and 64 staged slots


def sub_masks(x, y, a, b, c, o, X0, Y0, bw, bh):
    """block_mask16's rule (bounding box + Q-norm triangle inequality) for sub-blocks of bw x bh pixels: (n, 256/(bw bh)) bool."""
    vis = 255.0 * o
    det = a * c - b * b
    tau2 = 2.0 * (np.maximum(np.log(np.maximum(vis, 1e-30)), 0.0) + 1e-3)
    hx = np.sqrt(tau2 * c / det) * 1.001 + 0.05
    hy = np.sqrt(tau2 * a / det) * 1.001 + 0.05
    ex_, ey_ = (bw - 1) / 2.0, (bh - 1) / 2.0                             # half extents of the pixel centres of a sub-block
    qd, qo = ex_ * ex_ * a + ey_ * ey_ * c, 2.0 * ex_ * ey_ * b
    rb = np.sqrt(np.maximum(qd + qo, qd - qo))
    lim2 = (np.sqrt(tau2) * 1.001 + rb + 1e-3) ** 2
    xl, xh, yl, yh = x - hx - X0, x + hx - X0, y - hy - Y0, y + hy - Y0
    nbx, nby = 16 // bw, 16 // bh
    m = np.zeros((x.size, nbx * nby), dtype=bool)
    for by in range(nby):
        dy = (Y0 + bh * by + ey_) - y
        rowok = (yh >= bh * by) & (yl <= bh * by + bh - 1)
        for bx in range(nbx):
            dx = (X0 + bw * bx + ex_) - x
            Q = dx * (a * dx + 2.0 * b * dy) + c * dy * dy
            m[:, by * nbx + bx] = rowok & (xh >= bw * bx) & (xl <= bw * bx + bw - 1) & (Q <= lim2)
    m[~((det > 0) & (a > 0) & (c > 0))] = True
    m[~(vis >= 0.99)] = False
    return m


def contributing(x, y, a, b, c, o, X0, Y0):
    """(n, 256) bool: alpha >= 1/255 and power <= 0 at the pixel (the per-pixel test; the n_contrib cut is not modelled)."""
    px = X0[:, None] + (np.arange(256) % 16)[None, :]
    py = Y0[:, None] + (np.arange(256) // 16)[None, :]
    dx, dy = x[:, None] - px, y[:, None] - py
    power = -0.5 * (a[:, None] * dx * dx + c[:, None] * dy * dy) - b[:, None] * dx * dy
    return (power <= 0) & (o[:, None] * np.exp(power) >= 1.0 / 255.0)


def lists_per_chunk(mask, tile, ck, T):
    nck = int(ck.max()) + 1
    L = np.zeros((T, nck, mask.shape[1]), dtype=np.int64)
    for b in range(mask.shape[1]):
        sel = mask[:, b]
        np.add.at(L[:, :, b], (tile[sel], ck[sel]), 1)
    return L


def quadrant_groups(nbx, nby):
    """block indices of the four 8x8-pixel quadrants, for a grid of nbx x nby sub-blocks"""
    hx, hy = nbx // 2, nby // 2
    return [[(qy * hy + j) * nbx + qx * hx + i for j in range(hy) for i in range(hx)] for qy in range(2) for qx in range(2)]


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
    T = gx * gy
    _, pl, ranges, _ = O.bin_and_sort(pre.depth.float(), pre.rect, pre.tiles_touched, gx, gy, pre.keep)
    pl = pl.numpy().astype(np.int64)
    ranges = ranges.numpy().astype(np.int64)
    n_t = ranges[:, 1] - ranges[:, 0]
    I = pl.size
    tile = np.repeat(np.arange(T), n_t)
    pos = np.arange(I) - ranges[tile, 0]
    xy, conic, op = pre.xy.numpy().astype(np.float64), pre.conic.numpy().astype(np.float64), pre.opacity.numpy().astype(np.float64)
    X0, Y0 = (tile % gx) * 16.0, (tile // gx) * 16.0
    x, y, a, b, c, o = xy[pl, 0], xy[pl, 1], conic[pl, 0], conic[pl, 1], conic[pl, 2], op[pl]
    print(f"P={P} {W}x{H} sm={sm}: I={I}, instances/tile mean {n_t.mean():.0f}")

    # exact contribution per pixel, in batches (I x 256 booleans)
    geoms = {"4x4": (4, 4), "4x2": (4, 2), "2x2": (2, 2), "4x1": (4, 1), "8x2": (8, 2)}
    masks = {k: sub_masks(x, y, a, b, c, o, X0, Y0, *v) for k, v in geoms.items()}
    contrib_pairs = 0
    exact_entries = {k: 0 for k in geoms}
    missed = {k: 0 for k in geoms}
    pix_block = {k: ((np.arange(256) // 16) // v[1]) * (16 // v[0]) + ((np.arange(256) % 16) // v[0]) for k, v in geoms.items()}
    B = 100000
    for s0 in range(0, I, B):
        sl = slice(s0, min(I, s0 + B))
        cb = contributing(x[sl], y[sl], a[sl], b[sl], c[sl], o[sl], X0[sl], Y0[sl])
        contrib_pairs += int(cb.sum())
        for k in geoms:
            nb = masks[k].shape[1]
            ex = np.zeros((cb.shape[0], nb), dtype=bool)
            for blk in range(nb):
                ex[:, blk] = cb[:, pix_block[k] == blk].any(1)
            exact_entries[k] += int(ex.sum())
            missed[k] += int((ex & ~masks[k][sl]).sum())
    print(f"contributing pairs (alpha and power tests only) {contrib_pairs / 1e6:.1f} M")
    for k, v in geoms.items():
        ent = int(masks[k].sum())
        print(f"  sub-block {k}: entries {ent / 1e6:.2f} M ({ent / masks['4x4'].sum():.2f} x), evaluated pairs {ent * v[0] * v[1] / 1e6:.1f} M, "
              f"contributing / evaluated {contrib_pairs / (ent * v[0] * v[1]):.3f}, exact-reach entries {exact_entries[k] / 1e6:.2f} M, "
              f"reaching entries the rule misses: {missed[k]}")

    def price(name, L, groups, lanes, rows_per_entry, row_cost, step_overhead, slots_per_step, lists_per_wave, chunk):
        """L: (T, nck, nblocks); groups: list of block-index lists, one per wave; a wave runs max over its lists of ceil(L / lanes)"""
        steps = -(-L // lanes)
        wave_steps = np.stack([steps[:, :, g].max(2) for g in groups], 2)             # (T, nck, waves)
        live = int(L.sum())
        n_steps = int(wave_steps.sum())
        busy_chunks = int((np.stack([L[:, :, g].sum(2) for g in groups], 2) > 0).sum())
        cyc = n_steps * (rows_per_entry * row_cost + step_overhead) + busy_chunks * (CHUNK_OVERHEAD + LIST_ROUND * lists_per_wave * -(-chunk // 64))
        fill = live / max(1, n_steps * slots_per_step)
        quant = live / max(1, int((steps * lanes).sum()))
        print(f"  {name:58s} chunk {chunk:3d}: wave steps {n_steps / 1e3:7.1f} k, fill {fill:.3f} (quantisation only {quant:.3f}), "
              f"modelled {cyc / 1e6 / 1024:6.3f} ms on 1024 SIMDs")
        return cyc

    base = None
    for chunk in (176, 256, 352):
        ck = (n_t[tile] - 1 - pos) // chunk                                           # chunks are cut from the back of the list
        L44 = lists_per_chunk(masks["4x4"], tile, ck, T)
        L42 = lists_per_chunk(masks["4x2"], tile, ck, T)
        L22 = lists_per_chunk(masks["2x2"], tile, ck, T)
        q44, q42 = quadrant_groups(4, 4), quadrant_groups(4, 8)
        cA = price("A shipped: 4x4, 16-lane groups, wave = quadrant", L44, q44, 16, 4, ROW_STEP, STEP_OVERHEAD, 64, 4, chunk)
        if base is None:
            base = cA
        # E: compose + decompose = 8 packed per row step and pair of splats, one 16-lane scan for both
        row_e = 2 * ROW_STEP - 36 * C_DPP + 4 * C_DPP + 8 * C_PK
        price("E two splats per lane: 4x4, 32 entries per row group", L44, q44, 32, 4, row_e, 2 * STEP_OVERHEAD, 128, 4, chunk)
        halves = [list(range(0, 8)), list(range(8, 16))]
        price("B 4x4, 8-lane interleaved groups, wave = 8 blocks", L44, halves, 8, 4, ROW_STEP - SCAN_STEP, STEP_OVERHEAD + 4 * PHASE, 64, 8, chunk)
        price("C 4x2, 8-lane interleaved groups, wave = quadrant", L42, q42, 8, 2, ROW_STEP - SCAN_STEP, STEP_OVERHEAD + 4 * PHASE, 64, 8, chunk)
        d_groups = [[r * 4 + i for i in range(4)] for r in range(8)]                   # a wave = one row of four 4x2 sub-blocks
        price("D 4x2, 16-lane groups, wave = 4 sub-blocks (8 per tile)", L42, d_groups, 16, 2, ROW_STEP, STEP_OVERHEAD, 64, 4, chunk)
        f_groups = [[r * 4 + i for i in range(4)] for r in range(16)]
        price("F 2x2, 16-lane groups, wave = 4 sub-blocks (16 per tile)", L22, f_groups, 16, 1, ROW_STEP, STEP_OVERHEAD, 64, 4, chunk)
    # A with COMPACT per-wave copies: a wave's copy holds only the staged splats that reach its quadrant (S slots), so the same LDS stages a longer
    # chunk (C positions); a chunk is cut from the back of the list and ends at C positions or when some quadrant's copy is full.  LDS budget at
    # three workgroups per CU: 68 B per position (record, 16 list bytes, mask) + 4 x 48 B per slot <= 47 KB.
    m44 = masks["4x4"]
    quad_reach = np.stack([m44[:, g].any(1) for g in quadrant_groups(4, 4)], 1)             # (I, 4)
    print(f"  staged splats reaching a quadrant: {quad_reach.mean():.3f} of the positions")
    for C, S in ((176, 176), (256, 160), (288, 148), (320, 137), (384, 114)):
        ck = np.zeros(I, dtype=np.int64)
        for t in range(T):
            n = int(n_t[t])
            if n == 0:
                continue
            r0 = int(ranges[t, 0])
            qr = quad_reach[r0:r0 + n][::-1]                                                  # back to front
            cs = np.cumsum(qr, 0)
            k, start, out = 0, 0, np.empty(n, dtype=np.int64)
            while start < n:
                base_ = cs[start - 1] if start > 0 else 0
                end = min(n, start + C)
                over = np.nonzero(((cs[start:end] - base_) > S).any(1))[0]
                if over.size:
                    end = start + int(over[0])
                out[start:end] = k
                k += 1
                start = end
            ck[r0:r0 + n] = out[::-1]
        L = lists_per_chunk(m44, tile, ck, T)
        nchunks = int((L.sum(2) > 0).sum())
        price(f"A compact copies: <= {C} positions, {S} slots per wave ({nchunks / 1e3:.1f} k chunks)", L, quadrant_groups(4, 4), 16, 4, ROW_STEP, STEP_OVERHEAD + 2 * C_PLAIN, 64, 4, C)
    # the same with a FIXED staging window of C positions and a fallback instead of a variable extent: a window in which some quadrant's count
    # exceeds S runs its group phase twice, once per half (each half has at most C / 2 <= S reaching splats)
    for C, S in ((256, 160), (256, 152), (256, 144), (240, 160), (224, 176)):
        ck = np.zeros(I, dtype=np.int64)
        n_split = n_win = 0
        for t in range(T):
            n = int(n_t[t])
            if n == 0:
                continue
            r0 = int(ranges[t, 0])
            qr = quad_reach[r0:r0 + n][::-1]
            out = np.empty(n, dtype=np.int64)
            k = 0
            for start in range(0, n, C):
                end = min(n, start + C)
                n_win += 1
                if (qr[start:end].sum(0) > S).any():
                    mid = start + C // 2
                    out[start:mid] = k
                    out[mid:end] = k + 1
                    k += 2
                    n_split += 1
                else:
                    out[start:end] = k
                    k += 1
            ck[r0:r0 + n] = out[::-1]
        L = lists_per_chunk(m44, tile, ck, T)
        price(f"A compact copies, window {C}, {S} slots, halves on overflow ({n_split} of {n_win} windows)", L, quadrant_groups(4, 4), 16, 4, ROW_STEP, STEP_OVERHEAD + 2 * C_PLAIN, 64, 4, C)
    print("(the shipped kernel, A at chunk 176, measures 0.345 ms)")


if __name__ == "__main__":
    main()
