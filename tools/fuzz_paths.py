"""Randomised differential run on the GPU box (not a test): the tile-binned lists against the radix path's on scenes the parity
tests do not have -- depth in bands / on one plane / a shell around the camera, splats from tiny to screen-filling, scenes
shrunk into a handful of tiles, both tile rules and depth keys.  Lists, ranges, outputs: bit-identical; gradients: bit-identical
with the default tile rule, to summation order with the stock one.  Prints the tag of every trial before it runs (a device fault
ends the process: the last tag names the scene) and a summary.  Usage: python tools/fuzz_paths.py [--trials 150] [--seed 7]"""
import argparse, json, math, os, sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
from parity import assert_report, compare, run_hip, tile_sort_paths
from scenes import make_case, rel_err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=150)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--oracle", action="store_true", help="small scenes (<= 2500 Gaussians, <= 160 x 120) against the CPU oracle instead "
                                                          "of the radix path: integer artefacts bit for bit, floats by the parity tolerances")
    ap.add_argument("--long", action="store_true", help="bias towards long tile lists: 3000 .. 60000 Gaussians in compact scenes")
    ap.add_argument("--dense", type=int, default=0, help="BagsBackwardArgs.dense_per_tile (instances per tile above which the backward "
                                                         "keeps a byte per record instead of zero records; 1 forces that mode, -1 the other)")
    ap.add_argument("--cross-dense", action="store_true", help="differential runs only: the tile-binned run in dense-scene mode, the radix run "
                                                               "without it -- the gradients must still be bit-identical")
    ap.add_argument("--only", default="", help="comma-separated trial numbers: every scene is still drawn (the generator's state is the same as in "
                                               "the full run) but only these are run -- to look at a failure of an earlier run again")
    args = ap.parse_args()
    only = {int(x) for x in args.only.split(",") if x.strip()}
    cut = 3000 if only else 300
    from bags_raster import rasterizer as _R
    _R.DENSE_PER_TILE = args.dense
    rng = torch.Generator().manual_seed(args.seed)
    U = lambda a, b: float(torch.empty(1).uniform_(a, b, generator=rng))
    I = lambda a, b: int(torch.randint(a, b, (1,), generator=rng))
    bad, soft = [], []
    paths_hit = {}                                           # sort path (csrc/tile_sort.h) -> lists that took it, over all trials
    for trial in range(args.trials):
        P = max(1, int(math.exp(U(math.log(3000.0) if args.long else 0.0, math.log(60000.0 if args.long else 40000.0)))))
        W, H = I(16, 640), I(16, 480)
        if args.oracle:
            P, W, H = min(P, 2500), min(W, 160), min(H, 120)
        sm = math.exp(U(math.log(0.2), math.log(10.0)))
        deg = I(0, 4)
        depth = ("uniform", "banded", "flat", "shell")[I(0, 4)]
        shrink = ((0.3, 0.05, 0.02) if args.long else (1.0, 0.3, 0.05))[I(0, 3)]
        kw = dict(tile_bounds="aabb" if I(0, 3) == 0 else "opacity", depth_key="distance" if I(0, 2) else "z")
        tag = dict(trial=trial, P=P, W=W, H=H, sm=round(sm, 3), deg=deg, depth=depth, shrink=shrink, **kw)
        print(json.dumps(tag), flush=True)
        scene, cam = make_case(P, W, H, sm, deg, seed=5000 + trial)
        xyz = scene["means3D"] * shrink
        if depth == "banded":
            nb = I(2, 6)
            xyz[:, 2] = (torch.randint(0, nb, (P,), generator=rng).float() / nb - 0.5) * 2.0 + 0.01 * torch.randn(P, generator=rng)
        elif depth == "flat":
            xyz[:, 2] = 0.25
        elif depth == "shell":                               # around the camera (which sits at z = -4 looking at the origin)
            d = torch.randn(P, 3, generator=rng); d = d / d.norm(dim=1, keepdim=True)
            xyz = d * torch.empty(P, 1).uniform_(0.25, 1.5, generator=rng) + torch.tensor([0.0, 0.0, -4.0])
        scene["means3D"] = xyz
        if I(0, 3) == 0:
            scene["opacities"] = scene["opacities"] * 0.05    # long lists that do not saturate
        g = torch.randn(3, H, W, generator=rng)
        if only and trial not in only:
            continue
        if args.oracle:
            try:
                rep = compare(scene, cam, deg, check_fp64=True, **kw)
                try:
                    assert_report(rep, grad_tol=3e-4, skip_zero=("campos",))
                    print(f"  I = {rep['num_rendered'][0]}: ok", flush=True)
                except AssertionError as e:
                    # Two things a small random scene does that the kernels are not responsible for (seed 311 of round 5 has both): the fp32
                    # and the fp64 oracle disagree with EACH OTHER by more than assert_report's 2e-3 cap (one flipped alpha / transmittance
                    # threshold weighs more among 2500 Gaussians than among 500 k), and the device's v_exp_f32 decides such a pair differently
                    # from the oracle's libm exp on a pixel or two (n_contrib).  Second look: at most two such pixels, and for the "other
                    # oracle" bound 1.5 x what the oracles differ by themselves; everything else as strict as before.  Reported apart
                    # ("threshold_pairs"), and the pytest that runs this mode bounds how many trials of its pinned seed may land there.
                    o = rep.get("oracle32_vs_64", {})
                    relaxed = {k: (3e-4, max(2e-3, 1.5 * float(v))) for k, v in o.items()}
                    try:
                        assert_report(rep, grad_tol=3e-4, skip_zero=("campos",), tol_override=relaxed, n_contrib_mismatch=2.5 / (W * H),
                                      threshold_pixels=2)
                        print(f"  I = {rep['num_rendered'][0]}: ok on second look (threshold pair): {str(e)[:120]}", flush=True)
                        soft.append(dict(tag, why=str(e)[:120]))
                    except AssertionError as e2:
                        print(f"  I = {rep['num_rendered'][0]}: MISMATCH {str(e2)[:cut]}", flush=True)
                        bad.append(dict(tag, why=str(e2)[:300]))
            except Exception as e:
                print(f"  EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
                bad.append(dict(tag, why=f"exception {type(e).__name__}"))
            continue
        try:
            if args.cross_dense:
                _R.DENSE_PER_TILE = 1
                # whatever the op allocates next comes out of blocks full of 0xFF (torch's caching allocator hands them back): the dense-scene
                # mode reads a few bytes of slack behind its byte map, and round 5 once let them leak into the last Gaussian's marks
                poison = [torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda") for n in (1 << 12, 1 << 16, 1 << 20, 1 << 24)]
                del poison
            o_a, g_a, v_a = run_hip(scene, cam, deg, g, binning="auto", **kw)
            if args.cross_dense:
                _R.DENSE_PER_TILE = -1
            o_r, g_r, v_r = run_hip(scene, cam, deg, g, binning="radix", **kw)
            why = None
            if v_a["num_rendered"] != v_r["num_rendered"]:
                why = "num_rendered"
            else:
                for k in ("point_list", "keys_sorted", "n_contrib", "tiles_touched", "rect"):
                    if not torch.equal(v_a[k], v_r[k]):
                        why = k; break
            if why is None and not torch.equal(v_a["ranges"][:, 1] - v_a["ranges"][:, 0], v_r["ranges"][:, 1] - v_r["ranges"][:, 0]):
                why = "ranges"
            if why is None:
                for j, (a, b) in enumerate(zip(o_a, o_r)):
                    if not torch.equal(a, b):
                        why = f"output {j}"; break
            if why is None:
                for k in g_a:
                    if g_a[k] is None:
                        continue
                    if kw["tile_bounds"] == "opacity":
                        if not torch.equal(g_a[k], g_r[k]):
                            d = (g_a[k] - g_r[k]).abs()
                            rows = d.reshape(d.shape[0], -1).amax(1).nonzero().flatten() if d.dim() > 1 else d.nonzero().flatten()
                            tt = v_a["tiles_touched"]
                            why = (f"grad {k}: max |diff| {float(d.max()):.3e} of {float(g_r[k].abs().max()):.3e}, {int(rows.numel())} rows, first "
                                   f"{rows[:6].tolist()} with tiles_touched {[int(tt[r]) for r in rows[:6].tolist()] if d.dim() > 1 and d.shape[0] == tt.shape[0] else '-'}")
                            break
                    elif rel_err(g_a[k], g_r[k]) > 3e-3:
                        why = f"grad {k} {rel_err(g_a[k], g_r[k]):.2e}"; break
            longest = int((v_r["ranges"][:, 1] - v_r["ranges"][:, 0]).max()) if v_r["ranges"].numel() else 0
            took = tile_sort_paths(v_r["keys_sorted"], v_r["ranges"])          # which sorts the tile-binned run went through
            for k, c in took.items():
                paths_hit[k] = paths_hit.get(k, 0) + c
            print(f"  I = {v_r['num_rendered']}, longest list {longest}, sort paths {took}: {'ok' if why is None else 'MISMATCH ' + why}", flush=True)
            if why is not None:
                bad.append(dict(tag, why=why))
        except Exception as e:                               # (an exception of the op, not a device fault)
            print(f"  EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
            bad.append(dict(tag, why=f"exception {type(e).__name__}"))
    print(json.dumps({"trials": args.trials, "failures": bad, "threshold_pairs": soft, "sort_paths": paths_hit}))


if __name__ == "__main__":
    main()
