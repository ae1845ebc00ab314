"""CPU restatement of the reference's image-space distortion resampling -- TEST INFRASTRUCTURE ONLY (tests/ import it).

PARITY PINNED: tests/test_golden_cpu.py checks it against tests/golden/resample.npz, produced by running the reference's
own ``apply_distortion`` / ``center_crop`` (utils/util_distortion.py:58-77,271-311) here.

numpy, float64, explicit loops over the four taps (no torch): upsample of the control flow as F.interpolate(bilinear,
align_corners=False) does it (source = (dst + 0.5) * in/out - 0.5, clamped at 0, second tap clamped to the edge), sampling
as F.grid_sample(bilinear, zeros, align_corners=True) does it (pixel = (g + 1) / 2 * (size - 1)), crop by integer indexing
((Hf - Hc) // 2, (Wf - Wc) // 2), mask = not (channel 0 == 0 and channel 1 == 0).  The backward is the explicit adjoint."""
import numpy as np


def _flow_taps(n_in, n_out, idx):
    s = np.maximum((idx + 0.5) * (np.float32(n_in) / np.float32(n_out)) - 0.5, 0.0)
    i0 = np.minimum(s.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    lam = s - i0
    return i0, i1, lam


def _prepare(image, ctrl, flow_hw, crop_hw):
    C, H, W = image.shape
    h, w = ctrl.shape[:2]
    Hf, Wf = flow_hw; Hc, Wc = crop_hw
    ys = (Hf - Hc) // 2 + np.arange(Hc); xs = (Wf - Wc) // 2 + np.arange(Wc)
    y0, y1, ly = _flow_taps(h, Hf, ys.astype(np.float64)); x0, x1, lx = _flow_taps(w, Wf, xs.astype(np.float64))
    wy = [(1 - ly)[:, None], ly[:, None]]; wx = [(1 - lx)[None, :], lx[None, :]]
    yi = [y0, y1]; xi = [x0, x1]
    c = ctrl.astype(np.float64)
    flow = sum(wy[a] * wx[b] * 1.0 * c[yi[a]][:, xi[b]].transpose(2, 0, 1) for a in (0, 1) for b in (0, 1))   # (2,Hc,Wc)
    ix = (flow[0] + 1) / 2 * (W - 1); iy = (flow[1] + 1) / 2 * (H - 1)
    fx0 = np.floor(ix); fy0 = np.floor(iy)
    return dict(C=C, H=H, W=W, h=h, w=w, Hc=Hc, Wc=Wc, yi=yi, xi=xi, wy=wy, wx=wx, fx=ix - fx0, fy=iy - fy0,
                x0=fx0.astype(np.int64), y0=fy0.astype(np.int64))


def forward(image, ctrl, flow_hw, crop_hw):
    p = _prepare(image, ctrl, flow_hw, crop_hw)
    img = image.astype(np.float64)
    out = np.zeros((p["C"], p["Hc"], p["Wc"]))
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = p["y0"] + dy, p["x0"] + dx
            ok = (yy >= 0) & (yy < p["H"]) & (xx >= 0) & (xx < p["W"])
            wgt = (p["fy"] if dy else 1 - p["fy"]) * (p["fx"] if dx else 1 - p["fx"])
            out += np.where(ok, wgt, 0.0)[None] * img[:, np.clip(yy, 0, p["H"] - 1), np.clip(xx, 0, p["W"] - 1)]
    second = out[1] if out.shape[0] > 1 else out[0]            # the reference indexes channels 0 and 1 (RGB images)
    mask = (~((out[0] == 0) & (second == 0)))[None].astype(np.float32)
    return out, mask


def backward(image, ctrl, flow_hw, crop_hw, cot):
    p = _prepare(image, ctrl, flow_hw, crop_hw)
    img = image.astype(np.float64); cot = cot.astype(np.float64)
    H, W = p["H"], p["W"]
    g_img = np.zeros_like(img)
    v = {}
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = p["y0"] + dy, p["x0"] + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            yc, xc = np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)
            v[dy, dx] = np.where(ok[None], img[:, yc, xc], 0.0)
            wgt = np.where(ok, (p["fy"] if dy else 1 - p["fy"]) * (p["fx"] if dx else 1 - p["fx"]), 0.0)
            for c in range(p["C"]):
                np.add.at(g_img[c], (yc, xc), wgt * cot[c])
    dix = (cot * ((v[0, 1] - v[0, 0]) * (1 - p["fy"]) + (v[1, 1] - v[1, 0]) * p["fy"])).sum(0)
    diy = (cot * ((v[1, 0] - v[0, 0]) * (1 - p["fx"]) + (v[1, 1] - v[0, 1]) * p["fx"])).sum(0)
    ggx, ggy = dix * 0.5 * (W - 1), diy * 0.5 * (H - 1)
    g_ctrl = np.zeros((p["h"], p["w"], 2))
    for a in (0, 1):
        for b in (0, 1):
            wgt = p["wy"][a] * p["wx"][b]
            Y = np.broadcast_to(p["yi"][a][:, None], wgt.shape); X = np.broadcast_to(p["xi"][b][None, :], wgt.shape)
            np.add.at(g_ctrl[..., 0], (Y, X), wgt * ggx)
            np.add.at(g_ctrl[..., 1], (Y, X), wgt * ggy)
    return g_img, g_ctrl
