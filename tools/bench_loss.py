"""Times the fused photometric loss (csrc/loss.hip) fwd+bwd at 1080p against (a) the separable PyTorch implementation in
bags_raster/loss.py and (b) a dense 11x11 depthwise-conv formulation (what utils/loss_utils.py:58-76 launches), and prints
one JSON line with the HBM roofline of the fused kernels.  Usage: python tools/bench_loss.py [--steps 50]"""
import argparse, json, os, sys, time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd")]
from bags_raster import loss as L


def dense_loss(a, b, lam=0.2):
    w1 = L._window(11, 1.5, a)
    w2 = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda x: F.conv2d(x.unsqueeze(0), w2, padding=5, groups=3)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1; s2 = conv(b * b) - mu2 * mu2; s12 = conv(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - lam) * (a - b).abs().mean() + lam * (1 - m.mean())


def timed(fn, a, b, steps, warmup=5):
    for _ in range(warmup):
        x = a.clone().requires_grad_(True); fn(x, b).backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xs = [a.clone().requires_grad_(True) for _ in range(steps)]
    e0.record()
    for x in xs:
        fn(x, b).backward()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    a = torch.rand(3, args.height, args.width, generator=g).cuda()
    b = (a + 0.1 * torch.randn(3, args.height, args.width, generator=g).cuda()).clamp(0, 1)
    n = a.numel()
    t_fused = timed(L.fused_photometric_loss, a, b, args.steps)
    t_sep = timed(L.photometric_loss, a, b, max(5, args.steps // 5))
    t_dense = timed(dense_loss, a, b, max(5, args.steps // 5))
    alg = 11 * n * 4                 # fwd: read a,b, write 3 maps; bwd: read 3 maps + a,b, write grad
    out = dict(metric="photometric loss fwd+bwd @%dx%d" % (args.width, args.height), ms_fused=t_fused, ms_torch_separable=t_sep,
               ms_torch_dense_11x11=t_dense, speedup_vs_dense=t_dense / t_fused,
               roofline=dict(bound="hbm", alg_bytes=alg, achieved=alg / (t_fused * 1e-3) / 1e9, peak=8000.0, unit="GB/s",
                             frac=alg / (t_fused * 1e-3) / 8e12,
                             note="wall time per fwd+bwd incl. the autograd glue (clone, two scalar ops, stack); kernel-only times are in the rocprof summary"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
