"""Experiment: V views of one step rendered on ONE stream, one after the other, against the same views spread round-robin
over TWO / THREE streams (a view's forward front-end -- small, latency-bound kernels -- overlaps another view's blend
kernels).  Run once per HOST_WAIT mode: "forward" (the count is read inside every forward: the host blocks once per view)
and "lazy" (nothing in the forward blocks; the count is read at backward entry)."""
import json, sys, time
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
from bags_raster import rasterizer as R
from bags_raster.synth import sphere_views
from bags_raster.sharding import GradAllReducer
dev = torch.device('cuda', 0)
P, W, H, V = 500000, 1920, 1080, 4
scene, _ = bench.build_case(P, W, H, 0.5, 0, dev)
cams = sphere_views(V, W, H, noise=0.05)
fns, params = [], None
for c in cams:
    f, p, _ = bench.make_step(scene, c, dev, leaves=params)
    params = params or p
    fns.append(f)
red = GradAllReducer(params)
res = {}
for mode in ("forward", "lazy"):
    R.HOST_WAIT = mode
    for nstreams in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        cur = torch.cuda.current_stream()
        def step():
            red.begin()
            for s in streams: s.wait_stream(cur)
            for k, f in enumerate(fns):
                with torch.cuda.stream(streams[k % nstreams]):
                    f(False)
            for s in streams: cur.wait_stream(s)
        for _ in range(40): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 40
        for _ in range(K): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        res[f"{mode}/{nstreams}"] = {"ms_per_step": round(dt * 1e3, 4), "ms_per_view": round(dt * 1e3 / V, 4), "gaussians_per_s": V * P / dt,
                                     "grad_checksum": float(params[0].grad.double().abs().sum())}
print(json.dumps(res, indent=1))
