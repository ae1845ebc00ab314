"""Soak run on the GPU box (not a test, not a benchmark): N training-like iterations over changing views of one scene -- render()
with the shs / shs_rest pair, fused loss, backward, an SGD nudge of every leaf so that instance counts and capacities drift --
checking as it goes: no exception, finite loss and gradients, the image bit-identical to the concatenated-SH call every K
iterations, and how often the speculative forward had to redo its second phase.  Every 100 iterations the splats grow 3.3 x for
50 iterations (8 x the instances: lists of thousands of entries per tile, beyond the speculative capacity), and the poses
drift until some cameras sit inside the scene (depth-clustered lists): that regime found the two-level sort's early exits
(csrc/tile_sort.h, round 4).  SOAK_LOG=1 prints the phase of every iteration to stderr (with HIP_LAUNCH_BLOCKING=1 that
tells which call a device fault belongs to).  Usage: python tools/soak.py [--iters 600] [--P 300000]"""
import argparse, json, os, sys, warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd")]
from bags_raster import loss as L
from bags_raster import rasterizer as R
from bags_raster.gaussians import GaussianBag
from bags_raster.render import render, PipelineParams
from bags_raster.synth import synth_scene, sphere_views


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--P", type=int, default=200_000)
    ap.add_argument("--check-every", type=int, default=50)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--tile-bounds", default="opacity", choices=("opacity", "aabb"))
    ap.add_argument("--depth-key", default="z", choices=("z", "distance"))
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--hybrid", action="store_true", help="Python-side SH colours (the reference's default colour path: colors_precomp)")
    ap.add_argument("--host-wait", default="forward", choices=("forward", "lazy"),
                    help="lazy: the count is read at the backward's entry; an overflow recovers with a warning (LAZY_RECOVER)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    W, H = args.width, args.height
    R.HOST_WAIT = args.host_wait
    R.LAZY_RECOVER = args.host_wait == "lazy"
    import bags_raster.render
    RR = sys.modules["bags_raster.render"]
    settings_cls = RR.GaussianRasterizationSettings
    extra = {"tile_bounds": args.tile_bounds}

    def settings_with(**over):                               # render() builds the settings: route the soak's switches into them
        return lambda **kw: settings_cls(**dict(kw, **extra, **over))
    RR.GaussianRasterizationSettings = settings_with()
    pc = GaussianBag.from_activated(synth_scene(args.P, args.seed, 0.7, 3), 3, device=dev)
    cams = sphere_views(40, W, H, noise=0.1, device=dev)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(args.seed + 1)).to(dev)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    redo = 0
    losses = []
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        for it in range(args.iters):
            cam = cams[(it * 7) % len(cams)]
            leaves = pc.leaves() + cam.pose_leaves()
            for t in leaves:
                t.grad = None
            pc.active_sh_degree = min(3, it // 60)                  # the degree ramps up as in training
            log = (lambda m: (print(f"it {it}: {m}", file=sys.stderr), sys.stderr.flush())) if os.environ.get("SOAK_LOG") else (lambda m: None)
            log("render")
            out = render(cam, pc, pipe, bg, 0.0, None, hybrid=args.hybrid, depth_key=args.depth_key)
            log(f"rendered, I = {R.LAST_NUM_RENDERED}")
            loss = L.fused_photometric_loss(out["render"], gt)
            log("loss done")
            loss.backward()
            log("backward done")
            if it % args.check_every in (0, 1):
                with torch.no_grad():
                    ref = render(cam, type("C", (), {"__getattr__": lambda s, k: None if k == "_features_rest" else getattr(pc, k)})(),
                                 pipe, bg, 0.0, None, hybrid=args.hybrid, depth_key=args.depth_key)["render"]
                    RR.GaussianRasterizationSettings = settings_with(binning="radix")          # ... and the radix path's lists
                    rad = render(cam, pc, pipe, bg, 0.0, None, hybrid=args.hybrid, depth_key=args.depth_key)["render"]
                    RR.GaussianRasterizationSettings = settings_with()
                assert torch.equal(ref, out["render"]), f"iteration {it}: split-SH image differs from the concatenated call"
                assert torch.equal(rad, out["render"]), f"iteration {it}: tile-binned image differs from the radix path's"
                bad = [i for i, t in enumerate(leaves) if t.grad is None or not bool(torch.isfinite(t.grad).all())]
                assert not bad, f"iteration {it}: non-finite or missing gradients for leaves {bad}"
                losses.append(float(loss))
                assert losses[-1] == losses[-1], "NaN loss"
            with torch.no_grad():                                    # drift: positions, sizes and poses move, lists grow and shrink
                s = 1.0 + 0.3 * torch.sin(torch.tensor(it / 23.0)).item()
                for t in pc.leaves():
                    t.add_(t.grad, alpha=-1e-2)
                pc._scaling.add_(0.02 * (s - 1.0))
                if it % 100 == 50:
                    pc._scaling.add_(1.2)                          # every splat 3.3x larger: ~8x the instances, beyond the capacity headroom
                if it % 100 == 0 and it:
                    pc._scaling.add_(-1.2)
                for t in cam.pose_leaves():
                    t.add_(t.grad.clamp(-1, 1), alpha=-1e-3)
        torch.cuda.synchronize()
        redo = sum(1 for w in wlist if "capacity" in str(w.message).lower() or "overflow" in str(w.message).lower())
        other = sorted({str(w.message)[:80] for w in wlist if not ("capacity" in str(w.message).lower() or "overflow" in str(w.message).lower())})
    print(json.dumps({"iters": args.iters, "P": args.P, "size": [W, H], "tile_bounds": args.tile_bounds, "depth_key": args.depth_key, "losses": losses, "speculation_redo_warnings": redo, "other_warnings": other,
                      "last_num_rendered": int(getattr(R, "LAST_NUM_RENDERED", 0))}))


if __name__ == "__main__":
    main()
