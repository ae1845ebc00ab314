"""Fused distortion resampling (csrc/resample.hip behind bags_resample_forward / bags_resample_backward) against the reference's
golden vectors, the numpy oracle and, at full size, the PyTorch pipeline the reference runs."""
import os

import numpy as np
import pytest
import torch

from bags_raster.distortion import resample_image, resample_image_torch
from oracle import resample_oracle as RO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _hip(image, ctrl, fhw, chw, cot):
    img = torch.from_numpy(image).to(DEV).requires_grad_(True)
    ctl = torch.from_numpy(ctrl).to(DEV).requires_grad_(True)
    out, mask = resample_image(img, ctl, fhw, chw)
    (out * torch.from_numpy(cot).to(DEV)).sum().backward()
    return out.detach().cpu().numpy(), mask.cpu().numpy(), img.grad.cpu().numpy(), ctl.grad.cpu().numpy()


def test_resample_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "resample.npz"))
    fhw, chw = tuple(int(v) for v in g["flow_hw"]), tuple(int(v) for v in g["crop_hw"])
    out, mask, gi, gc = _hip(g["image"], g["ctrl"], fhw, chw, g["cot"])
    np.testing.assert_allclose(out, g["out"], atol=2e-4)               # the reference's crop is a second grid_sample (1e-4 px)
    assert (mask != g["mask"]).mean() < 0.002
    np.testing.assert_allclose(gi, g["d_image"], atol=3e-4)
    np.testing.assert_allclose(gc, g["d_ctrl"], rtol=2e-3, atol=2e-3 * np.abs(g["d_ctrl"]).max())


@pytest.mark.parametrize("C,H,W,h,w,fhw,chw", [(3, 33, 47, 5, 6, (40, 52), (31, 45)), (1, 8, 8, 2, 2, (9, 9), (9, 9)),
                                               (3, 20, 30, 20, 30, (20, 30), (20, 30)), (4, 17, 5, 3, 7, (64, 64), (1, 1))])
def test_resample_matches_oracle(C, H, W, h, w, fhw, chw):
    rng = np.random.default_rng(C * 100 + H)
    image = rng.random((C, H, W), dtype=np.float32)
    gy, gx = np.meshgrid(np.linspace(-1.3, 1.3, h), np.linspace(-1.3, 1.3, w), indexing="ij")   # reaches outside: zero padding
    ctrl = (np.stack((gx, gy), -1) + 0.1 * rng.standard_normal((h, w, 2))).astype(np.float32)
    cot = rng.standard_normal((C,) + chw).astype(np.float32)
    out, mask, gi, gc = _hip(image, ctrl, fhw, chw, cot)
    o_w, m_w = RO.forward(image, ctrl, fhw, chw)
    gi_w, gc_w = RO.backward(image, ctrl, fhw, chw, cot)
    np.testing.assert_allclose(out, o_w, atol=5e-5)                    # fp32 sampling positions: ~1e-5 px
    assert (mask != m_w).mean() < 0.01
    np.testing.assert_allclose(gi, gi_w, atol=2e-4 * max(1.0, np.abs(gi_w).max()))
    np.testing.assert_allclose(gc, gc_w, atol=2e-3 * max(1e-6, np.abs(gc_w).max()))


def test_resample_full_size_against_pytorch_pipeline():
    """Rendered 1080p image, 68x120 control flow upsampled to 1188x2112, cropped to 1080x1920 (the shape of the reference's
    flow_scale = 1.1 setting): agrees with interpolate + grid_sample + center_crop on the GPU, forward and backward."""
    g = torch.Generator().manual_seed(3)
    H, W, h, w, fhw, chw = 1080, 1920, 68, 120, (1188, 2112), (1080, 1920)
    image = torch.rand(3, H, W, generator=g).to(DEV)
    gy, gx = torch.meshgrid(torch.linspace(-1.1, 1.1, h), torch.linspace(-1.1, 1.1, w), indexing="ij")
    ctrl = (torch.stack((gx, gy), -1) + 0.01 * torch.randn(h, w, 2, generator=g)).to(DEV)
    cot = torch.randn(3, *chw, generator=g).to(DEV)
    def run(fn, dt):
        img = image.to(dt).clone().requires_grad_(True); ctl = ctrl.to(dt).clone().requires_grad_(True)
        out, mask = fn(img, ctl, fhw, chw)
        (out * cot.to(dt)).sum().backward()
        return out.detach().double(), mask.double(), img.grad.double(), ctl.grad.double()
    hip, t32, t64 = run(resample_image, torch.float32), run(resample_image_torch, torch.float32), run(resample_image_torch, torch.float64)
    d = (hip[0] - t64[0]).abs()                                        # fp32 sampling positions near x = 1900: ~1e-4 px,
    assert d.max().item() < 1e-3 and d.mean().item() < 5e-5            # times the O(1)/px slope of a noise image
    assert d.max().item() <= 1.5 * (t32[0] - t64[0]).abs().max().item() + 1e-5
    # the mask is an exact == 0 test: pixels on the rim of the zero-padding region flip with the last bit of the position
    mm = lambda a: (a[1] != t64[1]).double().mean().item()
    assert mm(hip) <= 1.5 * mm(t32) + 1e-4 and mm(hip) < 5e-3, (mm(hip), mm(t32))
    # The float32 PyTorch pipeline is itself ~5e-3 away from float64 in dL/dflow on a noise image (its crop is a second
    # grid_sample whose integer grid is reproduced to 1e-4 px); the fused kernels must be at least as close to float64.
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for k in (2, 3):
        assert rel(hip[k], t64[k]) <= 1.5 * rel(t32[k], t64[k]) + 1e-4, (k, rel(hip[k], t64[k]), rel(t32[k], t64[k]))
    print("dL/dimage rel err vs float64: fused %.2e, pytorch32 %.2e; dL/dflow: fused %.2e, pytorch32 %.2e" %
          (rel(hip[2], t64[2]), rel(t32[2], t64[2]), rel(hip[3], t64[3]), rel(t32[3], t64[3])))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for fn, name in ((resample_image, "fused"), (resample_image_torch, "pytorch")):
        img = image.clone().requires_grad_(True); ctl = ctrl.clone().requires_grad_(True)
        fn(img, ctl, fhw, chw)[0].backward(cot); torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            img.grad = None; ctl.grad = None
            fn(img, ctl, fhw, chw)[0].backward(cot)
        e1.record(); torch.cuda.synchronize()
        print(f"resample fwd+bwd @1080p {name}: {e0.elapsed_time(e1) / 10:.3f} ms")


def test_resample_backward_is_bitwise_reproducible_and_handles_minification():
    """dL/dimage is accumulated per source tile in 64-bit fixed point (no global atomics): two runs must agree bit for bit.
    The second flow squeezes the whole 160x208 output into the centre of the image, so that far more than 32 output tiles
    sample one source tile: the per-tile lists overflow and the gather falls back to testing every output tile's box."""
    rng = np.random.default_rng(11)
    C, H, W, h, w, fhw, chw = 3, 96, 128, 6, 8, (176, 224), (160, 208)
    image = rng.random((C, H, W), dtype=np.float32)
    cot = rng.standard_normal((C,) + chw).astype(np.float32)
    for span in (1.05, 0.08):
        gy, gx = np.meshgrid(np.linspace(-span, span, h), np.linspace(-span, span, w), indexing="ij")
        ctrl = (np.stack((gx, gy), -1) + 0.02 * span * rng.standard_normal((h, w, 2))).astype(np.float32)
        a = _hip(image, ctrl, fhw, chw, cot)
        b = _hip(image, ctrl, fhw, chw, cot)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        gi_w, gc_w = RO.backward(image, ctrl, fhw, chw, cot)
        np.testing.assert_allclose(a[2], gi_w, atol=2e-4 * max(1.0, np.abs(gi_w).max()))
        np.testing.assert_allclose(a[3], gc_w, atol=2e-3 * max(1e-6, np.abs(gc_w).max()))
        if span < 0.5:
            assert (np.abs(gi_w).reshape(C, -1).sum(0) > 0).mean() < 0.05      # everything lands in a small patch of the image


def test_resample_rejects_bad_arguments():
    with pytest.raises(RuntimeError, match="GPU tensor"):
        resample_image(torch.rand(3, 8, 8), torch.rand(2, 2, 2), (8, 8), (8, 8))
    with pytest.raises(RuntimeError, match="exceeds"):
        resample_image(torch.rand(3, 8, 8, device=DEV), torch.rand(2, 2, 2, device=DEV), (8, 8), (9, 8))
