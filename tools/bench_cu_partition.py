"""Experiment (round 6): the V views of one optimisation step (BASELINE configs 4/5; the reference's cubemap step renders five per
iteration, utils/cubemap_utils.py:229,263-265) on streams that are each RESTRICTED TO A CU PARTITION (hipExtStreamCreateWithCUMask),
against the same views on one stream and on plain streams.

Why partitions: a view is ~75 % VALU-bound blend work (blend_fwd / blend_bwd fill every CU's LDS and registers) and ~25 % HBM- or
latency-bound work (K1, tile_prefix, emit, preprocess_bwd, pose_reduce, launch gaps).  On plain streams the second kind of one
view cannot co-reside with the first kind of another (round 3: -4 %): its workgroups find no CU with room.  A partition gives every
view CUs of its own, so the latency-bound stages of one view run BESIDE the blends of another instead of behind them.

Gradients accumulate in place (rasterizer.ACCUMULATE_IN_PLACE) through an AccumulationGate: the per-Gaussian halves of the
backwards run in view order whatever the streams do, so the sums must be bit-identical to the one-stream run (checked here).

usage (GPU box): python tools/bench_cu_partition.py [--steps 40] [--views 4] [--host-wait lazy,forward]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import bench  # noqa: E402
from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, _lib, rasterizer as R  # noqa: E402
from bags_raster.synth import sphere_views  # noqa: E402
from scenes import camera_tensors  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--P", type=int, default=500000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--sm", type=float, default=0.5)
ap.add_argument("--host-wait", default="lazy,forward")
ap.add_argument("--only", default="", help="comma-separated configuration names")
args = ap.parse_args()

dev = torch.device("cuda", 0)
P, W, H, V = args.P, args.width, args.height, args.views
scene, _ = bench.build_case(P, W, H, args.sm, 0, dev)
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
_lib.load()
hip = C.CDLL("libamdhip64.so.7")          # the runtime libbags_raster.so is linked against (already loaded: same instance)
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
N_CU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(pred):
    words = (N_CU + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(N_CU):
        if pred(i):
            mask[i >> 5] |= 1 << (i & 31)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def fwd(v):
    rast, m2, md, sh, cts = v
    for t in (m2, md, sh, *cts):
        t.grad = None
    return rast(means3D=lv["means3D"], means2D=m2, means2D_densify=md, shift_factors=sh, shs=lv["shs"], colors_precomp=None,
                opacities=lv["opacities"], scales=lv["scales"], rotations=lv["rotations"], cov3D_precomp=None)[0]


# name -> list of stream factories (None = the current stream).  Two spellings of every split, because how mask bit i maps to
# (XCD, CU) is what tools/ubench/cu_mask_probe.hip finds out: "lo/hi" = contiguous bit ranges, "ilv" = bit i by i % 8.
CONFIGS = {
    "seq": [None],
    "plain2": [lambda: torch.cuda.Stream(), lambda: torch.cuda.Stream()],
    "plain4": [lambda: torch.cuda.Stream() for _ in range(4)],
    "halves_lohi": [lambda: masked_stream(lambda i: i < N_CU // 2), lambda: masked_stream(lambda i: i >= N_CU // 2)],
    "halves_ilv": [lambda: masked_stream(lambda i: i % 8 < 4), lambda: masked_stream(lambda i: i % 8 >= 4)],
    "halves_evenodd": [lambda: masked_stream(lambda i: i % 2 == 0), lambda: masked_stream(lambda i: i % 2 == 1)],
    "quarters_lohi": [(lambda q: (lambda: masked_stream(lambda i: i * 4 // N_CU == q)))(q) for q in range(4)],
    "quarters_ilv": [(lambda q: (lambda: masked_stream(lambda i: (i % 8) // 2 == q)))(q) for q in range(4)],
    # three quarters + one quarter: the odd view out takes the small partition
    "big_small_ilv": [lambda: masked_stream(lambda i: i % 8 < 6), lambda: masked_stream(lambda i: i % 8 >= 6)],
}
only = [x for x in args.only.split(",") if x]
res = {}
ref = None
gate = R.AccumulationGate()
for mode in args.host_wait.split(","):
    R.HOST_WAIT = mode
    for name, factories in CONFIGS.items():
        if only and name not in only:
            continue
        try:
            cur = torch.cuda.current_stream()
            streams = [cur if f is None else f() for f in factories]
            n = len(streams)
            R.ACCUMULATE_IN_PLACE = True
            R.ACCUMULATION_GATE = gate if n > 1 else None

            def step():
                gate.reset()
                for p_ in leaves:
                    p_.grad = None
                for s in streams:
                    if s is not cur:
                        s.wait_stream(cur)
                for k, v in enumerate(views):
                    with torch.cuda.stream(streams[k % n]):
                        fwd(v).backward(cot)
                for s in streams:
                    if s is not cur:
                        cur.wait_stream(s)
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            grads = [p_.grad.detach().clone() for p_ in leaves] + [t.grad.detach().clone() for v in views for t in v[4]]
            if ref is None:
                ref = grads
            same = all(torch.equal(a, b) for a, b in zip(ref, grads))
            res[f"{mode}/{name}"] = {"ms_per_view": round(dt * 1e3 / V, 4), "gaussians_per_s_M": round(V * P / dt / 1e6, 1),
                                     "streams": n, "bit_identical_to_first": bool(same)}
        except Exception as e:                                # a configuration the runtime refuses is a result, not a crash
            res[f"{mode}/{name}"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        finally:
            R.ACCUMULATE_IN_PLACE = False
            R.ACCUMULATION_GATE = None
        print(f"{mode}/{name}", res[f"{mode}/{name}"], flush=True)
print(json.dumps(res))
