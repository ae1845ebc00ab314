"""Fused pose -> matrix chain (csrc/camera.hip behind bags_camera_forward / bags_camera_backward) against the PyTorch chain
of bags_raster/camera.py, which tests/test_golden_cpu.py pins to the reference's getProjectionMatrix and
quaternion_to_rotation_matrix (scene/cameras.py:356-381,399-416; utils/graphics_utils.py:83-107)."""
import pytest
import torch

from bags_raster import camera as cam

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _make(seed, device):
    g = torch.Generator().manual_seed(seed)
    R = cam.quaternion_to_rotation(torch.randn(4, generator=g))
    T = torch.randn(3, generator=g) + torch.tensor([0.0, 0.0, 4.0])
    c = cam.PoseCamera(R, T, 1.1, 0.7, 64, 48, device=device)
    with torch.no_grad():
        c.delta_quaternion.copy_(0.05 * torch.randn(4, generator=g))
        c.delta_translation.copy_(0.1 * torch.randn(3, 1, generator=g))
        c.learnable_fovx.add_(0.03); c.learnable_fovy.sub_(0.02)
    return c


@pytest.mark.parametrize("align", [False, True])
def test_fused_camera_chain_matches_pytorch_chain(align):
    g = torch.Generator().manual_seed(9)
    grot = cam.quaternion_to_rotation(torch.tensor([1.0, 0.02, -0.03, 0.01])) if align else None
    gscale = torch.tensor(1.3) if align else None
    cots = [torch.randn(4, 4, generator=g), torch.randn(4, 4, generator=g), torch.randn(4, 4, generator=g), torch.randn(3, generator=g)]
    # reference chain on the CPU in float64-free plain float32 PyTorch
    c0 = _make(4, "cpu")
    gr0 = None if grot is None else grot.clone().requires_grad_(True)
    gs0 = None if gscale is None else gscale.clone().requires_grad_(True)
    want = c0.get_matrices(gr0, gs0)
    loss = sum((w * k).sum() for w, k in zip(want, cots))
    leaves0 = c0.pose_leaves() + ([gr0, gs0] if align else [])
    gwant = torch.autograd.grad(loss, leaves0)
    # fused chain on the GPU
    c1 = _make(4, DEV)
    gr1 = None if grot is None else grot.to(DEV).requires_grad_(True)
    gs1 = None if gscale is None else gscale.to(DEV).requires_grad_(True)
    got = c1.get_matrices(gr1, gs1)
    for a, b in zip(got, want):
        assert torch.allclose(a.cpu(), b.detach(), rtol=1e-5, atol=2e-6), (a.cpu() - b.detach()).abs().max()
    loss = sum((w * k.to(DEV)).sum() for w, k in zip(got, cots))
    leaves1 = c1.pose_leaves() + ([gr1, gs1] if align else [])
    ggot = torch.autograd.grad(loss, leaves1)
    for a, b in zip(ggot, gwant):
        assert a.shape == b.shape
        assert torch.allclose(a.cpu(), b, rtol=2e-4, atol=2e-5), (a.cpu(), b)


def test_fused_camera_chain_partial_gradients_and_errors():
    c = _make(5, DEV)
    V, M, K, C = c.get_matrices()
    (gq,) = torch.autograd.grad(C.sum(), [c.delta_quaternion], retain_graph=True)      # only one leaf, only campos upstream
    assert torch.isfinite(gq).all() and gq.abs().sum() > 0
    (gf,) = torch.autograd.grad(K[0, 0], [c.learnable_fovx])
    t = torch.tan(c.learnable_fovx.detach() * 0.5)
    assert abs(gf.item() - (-(1 + t * t) / (2 * t * t)).item()) < 1e-4
    with pytest.raises(RuntimeError, match="GPU"):
        cam.fused_camera_chain(torch.zeros(4), torch.zeros(3), torch.tensor(1.0), torch.tensor(1.0), torch.tensor([1.0, 0, 0, 0]), torch.zeros(3))


def test_render_uses_fused_chain_and_reaches_pose_leaves():
    """render() -> PoseCamera.get_matrices -> HIP camera chain -> rasterizer; gradients arrive on the four pose leaves and
    agree with the same render driven by the PyTorch chain."""
    from bags_raster.gaussians import GaussianBag
    from bags_raster.render import render, PipelineParams
    from bags_raster.synth import synth_scene, sphere_views
    scene = synth_scene(1200, 5, 1.5, 3)
    W, H = 128, 96
    gimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(DEV)

    class Plain:                                   # hides get_matrices: render() falls back to the four getters
        def __init__(self, c): self.c = c
        def __getattr__(self, k):
            if k == "get_matrices":
                raise AttributeError(k)
            return getattr(self.c, k)
    res = []
    for wrap in (lambda c: c, Plain):
        c = sphere_views(2, W, H, noise=0.05, device=DEV)[1]
        pc = GaussianBag.from_activated(scene, 3, device=DEV)
        out = render(wrap(c), pc, PipelineParams(), torch.zeros(3, device=DEV), 0.0, None, hybrid=False)
        out["render"].backward(gimg)
        res.append((out["render"].detach().cpu(), [p.grad.detach().cpu().clone() for p in c.pose_leaves()]))
    assert (res[0][0] - res[1][0]).abs().max().item() < 2e-4
    for a, b in zip(res[0][1], res[1][1]):
        assert (a - b).norm().item() <= 2e-3 * b.norm().item() + 1e-6, (a, b)


def test_fused_camera_chain_matches_reference_methods(golden_dir):
    """csrc/camera.hip against tests/golden/camera_pose_chain.npz: the values and Jacobians the reference's own Camera methods
    (scene/cameras.py:356-381, run on a stub self by make_golden.py) produce, with and without global alignment."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "camera_pose_chain.npz"))
    rs = np.random.RandomState(0)
    for i in range(g["x"].shape[0]):
        x = torch.tensor(g["x"][i], device=DEV)
        leaves = [x[0:4].clone().requires_grad_(True), x[4:7].clone().view(3, 1).requires_grad_(True), x[7].clone().requires_grad_(True),
                  x[8].clone().requires_grad_(True), x[9:18].clone().view(3, 3).requires_grad_(True), x[18:19].clone().requires_grad_(True)]
        q0 = torch.tensor(g["init_quaternion"][i], device=DEV); t0 = torch.tensor(g["init_translation"][i], device=DEV)
        V, M, K, C = cam.fused_camera_chain(leaves[0], leaves[1], leaves[2], leaves[3], q0, t0, 0.01, 100.0, leaves[4], leaves[5])
        y = torch.cat([V.reshape(-1), M.reshape(-1), K.reshape(-1), C.reshape(-1)])
        assert np.allclose(y.detach().cpu().numpy(), g["y"][i], rtol=2e-5, atol=2e-6), np.abs(y.detach().cpu().numpy() - g["y"][i]).max()
        J = g["dy_dx"][i].astype(np.float64)
        for _ in range(4):                                    # vector-Jacobian products with random cotangents
            c = rs.randn(51)
            gs = torch.autograd.grad(y, leaves, torch.tensor(c, dtype=torch.float32, device=DEV), retain_graph=True)
            got = torch.cat([a.reshape(-1) for a in gs]).cpu().numpy().astype(np.float64)
            want = c @ J
            assert np.abs(got - want).max() <= 3e-4 * np.abs(want).max() + 1e-5, (i, np.abs(got - want).max(), np.abs(want).max())
        if i < 2:                                             # no alignment: the None / None call gives the same four tensors
            V0, M0, K0, C0 = cam.fused_camera_chain(leaves[0], leaves[1], leaves[2], leaves[3], q0, t0)
            y0 = torch.cat([V0.reshape(-1), M0.reshape(-1), K0.reshape(-1), C0.reshape(-1)])
            assert np.allclose(y0.detach().cpu().numpy(), g["y"][i], rtol=2e-5, atol=2e-6)
