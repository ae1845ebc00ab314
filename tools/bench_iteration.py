"""One optimisation iteration as train.py runs it around the op (camera chain -> render -> photometric loss -> backward),
BASELINE config 3 shape (500 k Gaussians, 1920x1080, SH 3, pose leaves learnable), timed two ways:
  fused      PoseCamera.get_matrices (csrc/camera.hip) + rasterizer + fused_photometric_loss (csrc/loss.hip)
  unfused    the four camera getters in PyTorch + rasterizer + dense 11x11 depthwise-conv SSIM (what utils/loss_utils.py launches)
The rasterizer is the same HIP library in both; the difference is the neighbours SURVEY.md section 8(f) ranks 1 and 4 name.
Prints one JSON line.  Usage: python tools/bench_iteration.py [--steps 20]"""
import argparse, json, os, sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd")]
from bags_raster import loss as L
from bags_raster.gaussians import GaussianBag
from bags_raster.render import render, PipelineParams
from bags_raster.synth import synth_scene, sphere_views


def dense_loss(a, b, lam=0.2):
    w1 = L._window(11, 1.5, a)
    w2 = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda x: F.conv2d(x.unsqueeze(0), w2, padding=5, groups=3)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1; s2 = conv(b * b) - mu2 * mu2; s12 = conv(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - lam) * (a - b).abs().mean() + lam * (1 - m.mean())


class ConcatOnly:
    """Hides the container's fused activations' features=False form, so that render() passes get_features (the concatenation)."""
    def __init__(self, pc): self.pc = pc
    def __getattr__(self, k):
        if k == "_features_rest":
            return None
        return getattr(self.pc, k)


class PlainCamera:
    """Hides get_matrices so that render() evaluates the four PyTorch getters, as the reference does."""
    def __init__(self, c): self.c = c
    def __getattr__(self, k):
        if k == "get_matrices":
            raise AttributeError(k)
        return getattr(self.c, k)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--fused-only", action="store_true", help="only the fused leg (for kernel traces)")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of the fused iteration's HOST side (lazy host wait, so that "
                                                                "nothing blocks): where the Python time per iteration goes")
    ap.add_argument("--unpacked", action="store_true", help="get_features (torch.cat of features_dc and features_rest, as the reference "
                                                            "calls the op) instead of the shs / shs_rest pair"); args = ap.parse_args()
    dev = torch.device("cuda", 0)
    P, W, H = 500_000, 1920, 1080
    scene = synth_scene(P, 0, 0.5, 3)
    cam = sphere_views(1, W, H, noise=0.05, device=dev)[0]
    pc = GaussianBag.from_activated(scene, 3, device=dev)
    if args.unpacked:                                   # hide the pair: render() then concatenates, as the reference does
        pc = ConcatOnly(pc)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    leaves = pc.leaves() + cam.pose_leaves()

    def iteration(camera, loss_fn):
        for t in leaves:
            t.grad = None
        out = render(camera, pc, pipe, bg, 0.0, None, hybrid=False)
        loss_fn(out["render"], gt).backward()

    def timed(camera, loss_fn):
        for _ in range(3):
            iteration(camera, loss_fn)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            iteration(camera, loss_fn)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps
    if args.host_profile:
        import cProfile, pstats, time
        from bags_raster import rasterizer as R
        R.HOST_WAIT = "lazy"
        for _ in range(20):
            iteration(cam, L.fused_photometric_loss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            iteration(cam, L.fused_photometric_loss)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"host enqueue {1e3 * (t1 - t0) / 200:.3f} ms/iteration, with the device {1e3 * (t2 - t0) / 200:.3f} ms/iteration")
        pr = cProfile.Profile(); pr.enable()
        for _ in range(100):
            iteration(cam, L.fused_photometric_loss)
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        return
    t_fused = timed(cam, L.fused_photometric_loss)
    if args.fused_only:
        print(json.dumps({"fused": t_fused, "split_sh": not args.unpacked})); return
    t_unfused = timed(PlainCamera(cam), dense_loss)
    t_mixed = timed(PlainCamera(cam), L.fused_photometric_loss)
    print(json.dumps({"metric": "ms per iteration (camera chain + render + loss + backward) @1920x1080, 500k Gaussians",
                      "fused": t_fused, "pytorch_camera_chain_fused_loss": t_mixed, "pytorch_camera_chain_dense_ssim": t_unfused,
                      "speedup": t_unfused / t_fused, "split_sh": not args.unpacked}))


if __name__ == "__main__":
    main()
