"""Probe (diagnostic, not product): what does an HBM-bound kernel of ~preprocess_fwd's colour half cost when it runs
(a) on the op's stream, in series, or (b) on a second stream beside the forward's emission + per-tile sorts?
The difference is what splitting preprocess_fwd into a geometry kernel (needed by the binning chain) and a colour kernel
(needed only by blend_fwd) and running the latter on a side stream could buy.

The side kernel is a plain device copy of `MB` megabytes (read + write = 2 x MB of traffic), forked behind the prepare phase
(an event between bags_forward_prepare_async and bags_forward_finish_speculative) and joined at the end of the step.
usage: python tools/overlap_probe.py [MB]"""
import sys, time
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
from bags_raster import _lib as L

MB = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
dev = torch.device('cuda', 0)
scene, cam = bench.build_case(500000, 1920, 1080, 0.5, 0, dev)
step, params, ct = bench.make_step(scene, cam, dev)
lib = L.load()
n = int(MB * 1e6 / 4)
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
side = torch.cuda.Stream(dev)
ev = torch.cuda.Event()
orig = lib.bags_forward_finish_speculative
mode = {"m": "none"}


def patched(*a):
    if mode["m"] == "serial":
        y.copy_(x)
    elif mode["m"] == "side":
        main = torch.cuda.current_stream(dev)
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            y.copy_(x)
    return orig(*a)


lib.bags_forward_finish_speculative = patched


def run(m, N=300):
    mode["m"] = m
    main = torch.cuda.current_stream(dev)
    for _ in range(60):
        step()
        if m == "side":
            main.wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
        if m == "side":
            main.wait_stream(side)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


# the copy alone
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20):
    y.copy_(x)
e0.record()
for _ in range(100):
    y.copy_(x)
e1.record()
torch.cuda.synchronize()
print(f"copy of {MB:.0f} MB alone: {e0.elapsed_time(e1) * 10:.1f} us")
for rep in range(2):
    a, b, c = run("none"), run("serial"), run("side")
    print(f"ms/step: no extra kernel {a:.4f}, in series {b:.4f} (+{1e3 * (b - a):.1f} us), on a side stream beside emit + sort {c:.4f} "
          f"(+{1e3 * (c - a):.1f} us)")
