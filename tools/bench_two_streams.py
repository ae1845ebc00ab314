"""Experiment: the V views of one iteration (BASELINE configs 4/5, the cubemap step utils/cubemap_utils.py:229,263-265) on ONE
stream against the same views spread round-robin over TWO / THREE streams, so that one view's small latency-bound kernels
(preprocess, binning, preprocess_bwd) overlap another view's VALU-bound blend kernels.

  order "per_view": forward + backward of view 0, then of view 1, ...            (what bench.py's V-view leg does)
  order "batched":  all V forwards, then all V backwards                          (render the batch, then differentiate it)
  HOST_WAIT "forward": the instance count is read inside every forward (the host blocks once per view)
  HOST_WAIT "lazy":    nothing in a forward blocks; the count is read at the entry of that view's backward
"""
import json, math, sys, time
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, rasterizer as R
from bags_raster.synth import sphere_views
from bags_raster.sharding import GradAllReducer
from scenes import camera_tensors
dev = torch.device('cuda', 0)
P, W, H, V = 500000, 1920, 1080, 4
scene, _ = bench.build_case(P, W, H, 0.5, 0, dev)
cams = sphere_views(V, W, H, noise=0.05)
leaves = [v.clone().requires_grad_(True) for v in scene.values()]
lv = dict(zip(scene.keys(), leaves))
cot = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
views = []
for c in cams:
    ct = {k: v.clone().requires_grad_(True) for k, v in camera_tensors(c, dev).items()}
    m2, md, sh = (torch.zeros(P, 3, device=dev, requires_grad=True), torch.zeros(P, 3, device=dev, requires_grad=True),
                  torch.zeros(3, device=dev, requires_grad=True))
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(c.FoVx * 0.5), tanfovy=math.tan(c.FoVy * 0.5),
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                       projmatrix=ct["projmatrix"], intrinsic=ct["intrinsic"], sh_degree=3, campos=ct["campos"],
                                       prefiltered=False, debug=False, debug_iter=0)
    views.append((GaussianRasterizer(st), m2, md, sh, list(ct.values())))


def fwd(v):
    rast, m2, md, sh, cts = v
    for t in (m2, md, sh, *cts):
        t.grad = None
    return rast(means3D=lv["means3D"], means2D=m2, means2D_densify=md, shift_factors=sh, shs=lv["shs"], colors_precomp=None,
                opacities=lv["opacities"], scales=lv["scales"], rotations=lv["rotations"], cov3D_precomp=None)[0]


red = GradAllReducer(leaves)
res = {}
import os
MODES = os.environ.get("MODES", "forward,lazy").split(",")
for mode in MODES:
    R.HOST_WAIT = mode.split(":")[0]
    if ":" in mode:
        R.CAPACITY_HEADROOM = float(mode.split(":")[1])
    for order in ("per_view", "batched"):
        for nstreams in (1, 2, 3):
            streams = [torch.cuda.Stream() for _ in range(nstreams)]
            cur = torch.cuda.current_stream()

            def step():
                red.begin()
                for s in streams: s.wait_stream(cur)
                if order == "per_view":
                    for k, v in enumerate(views):
                        with torch.cuda.stream(streams[k % nstreams]):
                            fwd(v).backward(cot)
                else:
                    imgs = []
                    for k, v in enumerate(views):
                        with torch.cuda.stream(streams[k % nstreams]):
                            imgs.append(fwd(v))
                    for k, im in enumerate(imgs):
                        with torch.cuda.stream(streams[k % nstreams]):
                            im.backward(cot)
                for s in streams: cur.wait_stream(s)
            for _ in range(40): step()
            torch.cuda.synchronize()
            seg0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
            t0 = time.perf_counter()
            K = 40
            for _ in range(K): step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
            res[f"{mode}/{order}/{nstreams}"] = {"ms_per_view": round(dt * 1e3 / V, 4), "gaussians_per_s": round(V * P / dt / 1e6, 1),
                                                 "device_allocs_in_timed_region": torch.cuda.memory_stats().get("num_device_alloc", 0) - seg0,
                                                 "grad_checksum": float(leaves[0].grad.double().abs().sum())}
            print(f"{mode}/{order}/{nstreams}", res[f"{mode}/{order}/{nstreams}"], flush=True)
print(json.dumps(res))
