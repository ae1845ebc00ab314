"""How long the HOST needs to enqueue one forward + backward of the bench workload (no waiting for the device), against the
device's time per step: the margin by which the Python side stays ahead of the GPU."""
import sys, time
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd', 'tests']
import torch, bench
import cProfile, pstats
dev = torch.device('cuda', 0)
scene, cam = bench.build_case(500000, 1920, 1080, 0.5, 0, dev)
step, params, ct = bench.make_step(scene, cam, dev)
for _ in range(50): step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, device-bound total {1e3 * (t2 - t0) / N:.3f} ms/step")
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
