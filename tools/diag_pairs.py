"""Diagnostic (DIAG_PAIRS build only, tools/diag_pairs.sh): evaluated vs contributing (pixel, splat) pairs of the blend
kernels on the bench workload -> gpurun_out/pairs.json (copied to profiles/ and read by bench.py for roofline_valu)."""
import ctypes as C, json, sys
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
from bags_raster import _lib, rasterizer as R
P, W, H, sm = 500000, 1920, 1080, 0.5
res = {"key": {"P": P, "width": W, "height": H, "sm": sm}}
dev = torch.device('cuda', 0)
lib = _lib.load()
out = (C.c_ulonglong * 8)()
scene, cam = bench.build_case(P, W, H, sm, 0, dev)
for tb in ("opacity", "aabb"):
    step, params, ct = bench.make_step(scene, cam, dev, tile_bounds=tb)
    step(); torch.cuda.synchronize()
    lib.bags_diag_pairs(out, 1)
    n = 3
    for _ in range(n): step()
    torch.cuda.synchronize()
    lib.bags_diag_pairs(out, 1)
    v = [x / n for x in out]
    res[tb] = {"instances_I": int(R.LAST_NUM_RENDERED), "bwd_pairs_evaluated": v[0], "bwd_pairs_contributing": v[1],
               "fwd_pairs_evaluated": v[2], "fwd_pairs_contributing": v[3],
               "bwd_entries": v[4], "bwd_entries_without_contribution": v[5], "bwd_wave_steps": v[6], "bwd_wave_chunks": v[7]}
print(json.dumps(res))
