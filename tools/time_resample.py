import sys, torch
sys.path[:0] = ['.', 'bundle-adjusting-gaussian-splatting_amd']
from bags_raster.distortion import resample_image
g = torch.Generator().manual_seed(3)
H, W, h, w, fhw, chw = 1080, 1920, 68, 120, (1188, 2112), (1080, 1920)
image = torch.rand(3, H, W, generator=g).cuda()
gy, gx = torch.meshgrid(torch.linspace(-1.1, 1.1, h), torch.linspace(-1.1, 1.1, w), indexing="ij")
ctrl = (torch.stack((gx, gy), -1) + 0.01 * torch.randn(h, w, 2, generator=g)).cuda()
cot = torch.randn(3, *chw, generator=g).cuda()
def t(ri, rc, bwd=True):
    img = image.clone().requires_grad_(ri); ctl = ctrl.clone().requires_grad_(rc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        o = resample_image(img, ctl, fhw, chw)[0]
        if bwd: o.backward(cot)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        o = resample_image(img, ctl, fhw, chw)[0]
        if bwd: o.backward(cot)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
print("fwd only %.3f | +bwd image %.3f | +bwd ctrl %.3f | +bwd both %.3f ms" % (t(True, True, False), t(True, False), t(False, True), t(True, True)))
