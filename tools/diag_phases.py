"""Diagnostic (DIAG_PHASES build only): share of wave cycles per phase of blend_bwd_scan_kernel."""
import ctypes as C, subprocess, sys, os
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
from bags_raster import _lib
dev = torch.device('cuda', 0)
scene, cam = bench.build_case(500000, 1920, 1080, 0.5, 0, dev)
step, params, ct = bench.make_step(scene, cam, dev)
for _ in range(3): step()
torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_ulonglong * 8)()
lib.bags_diag_phases(out)
v = list(out); tot = sum(v)
names = ['prologue', 'staging', 'barrier1', 'lists', 'groups', 'barrier2', 'write', '-']
print({n: round(x / tot, 3) for n, x in zip(names, v)}, 'total cycles per launch', tot / 3)
