"""Diagnostic: accuracy of dL/dshift_factors (and the other pose gradients) of the HIP path against both oracles."""
import sys
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch
from parity import compare
from scenes import make_case
torch.set_num_threads(16)
for (P, W, H, sm, deg, sf) in [(1500, 128, 96, 2.0, 2, [0.05, -0.02, 0.01]), (20000, 400, 304, 1.0, 3, [-0.03, 0.02, 0.015]),
                               (20000, 400, 304, 1.0, 3, [0, 0, 0]), (120000, 960, 544, 0.7, 3, [0.02, -0.01, 0.005])]:
    scene, cam = make_case(P, W, H, sm, deg, seed=9)
    rep = compare(scene, cam, deg, shift=torch.tensor(sf, dtype=torch.float32))
    keys = ("shift_factors", "viewmatrix", "projmatrix", "intrinsic", "campos", "means3D")
    print(P, sf, "ints", all(rep[k] for k in ("radii_equal", "rect_equal", "depth_bits_equal", "point_list_equal", "keys_equal", "ranges_equal")),
          "I", rep["num_rendered"], "img", "%.1e" % rep["image_max_err"],
          "hip-vs-32", {k: "%.1e" % rep["grad_rel_fp32"][k] for k in keys},
          "hip-vs-64", {k: "%.1e" % rep["grad_rel_fp64"][k] for k in keys},
          "32-vs-64", {k: "%.1e" % rep["oracle32_vs_64"][k] for k in keys}, flush=True)
