import sys, torch
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd']
from bags_raster.gaussians import GaussianBag
from bags_raster.synth import synth_scene
pc = GaussianBag.from_activated(synth_scene(500000, 0, 0.5, 3), 3, device="cuda")
def step():
    for t in pc.leaves(): t.grad = None
    outs = [pc.get_xyz, pc.get_features, pc.get_opacity, pc.get_scaling, pc.get_rotation]
    torch.autograd.backward(outs[1:], [torch.ones_like(o) for o in outs[1:]])
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print("activations fwd+bwd (PyTorch): %.3f ms" % (e0.elapsed_time(e1) / 20))
