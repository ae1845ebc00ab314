"""The op's backward hands autograd views of ONE buffer for the replicated Gaussian parameters, so that the view-sharded
exchange (bags_raster/sharding.py) is a single collective.  This must survive autograd's gradient accumulation."""
import pytest
import torch

from scenes import hip_settings, make_case

pytestmark = pytest.mark.gpu


def test_gaussian_gradients_share_one_storage_and_coalesce():
    from bags_raster import GaussianRasterizer
    from bags_raster.sharding import coalesce_by_storage
    dev = torch.device("cuda", 0)
    scene, cam = make_case(3000, 160, 96, 1.5, 3, seed=2)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
    P = leaves["means3D"].shape[0]
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, 3, dev))
    img = rast(means3D=leaves["means3D"], means2D=m2d, means2D_densify=None, shift_factors=None, shs=leaves["shs"],
               colors_precomp=None, opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
               cov3D_precomp=None)[0]
    img.sum().backward()
    grads = [leaves[k].grad for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert len({g.untyped_storage().data_ptr() for g in grads}) == 1, "autograd cloned the carved gradients"
    merged = coalesce_by_storage(grads + [m2d.grad])
    assert len(merged) == 2                                    # the flat buffer + the means2D gradient (a tensor of its own)
    flat = max(merged, key=lambda t: t.numel())
    assert flat.numel() >= sum(g.numel() for g in grads)
    # summing "over one rank" through the flat view must be the identity on every gradient
    before = [g.clone() for g in grads]
    flat.mul_(2.0)
    for g, b in zip(grads, before):
        assert torch.equal(g, 2.0 * b)
    # a second backward accumulates into the same tensors (no re-carving needed for correctness)
    img2 = rast(means3D=leaves["means3D"], means2D=m2d, means2D_densify=None, shift_factors=None, shs=leaves["shs"],
                colors_precomp=None, opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                cov3D_precomp=None)[0]
    img2.sum().backward()
    for g, b in zip(grads, before):
        assert torch.allclose(g, 3.0 * b, rtol=1e-5, atol=1e-6)
