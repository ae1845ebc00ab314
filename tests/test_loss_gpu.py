"""Fused photometric loss (csrc/loss.hip behind bags_loss_forward / bags_loss_backward) against the oracle, the reference's
golden vectors and, at the bench size, the separable PyTorch implementation."""
import os

import numpy as np
import pytest
import torch

from bags_raster import loss as L
from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fused(a, b, g_l1, g_ssim):
    at = torch.from_numpy(a).to(DEV).requires_grad_(True)
    bt = torch.from_numpy(b).to(DEV)
    l1, s = L.fused_l1_ssim(at, bt)
    (g_l1 * l1 + g_ssim * s).backward()
    return l1.item(), s.item(), at.grad.cpu().numpy()


def test_fused_loss_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    l1, s, grad = _fused(g["a"], g["b"], 0.8, -0.2)
    assert abs(l1 - float(g["l1"])) < 1e-6 and abs(s - float(g["ssim"])) < 1e-5
    np.testing.assert_allclose(grad, g["dloss_da"], rtol=1e-3, atol=2e-8)
    at = torch.from_numpy(g["a"]).to(DEV); bt = torch.from_numpy(g["b"]).to(DEV)
    assert abs(L.fused_photometric_loss(at, bt).item() - float(g["loss"])) < 1e-5
    g = np.load(os.path.join(golden_dir, "loss_odd.npz"))
    l1, s, g1 = _fused(g["a"], g["b"], 1.0, 0.0)
    _, _, g2 = _fused(g["a"], g["b"], 0.0, 1.0)
    assert abs(l1 - float(g["l1"])) < 1e-6 and abs(s - float(g["ssim"])) < 1e-5
    np.testing.assert_allclose(g1, g["dl1_da"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(g2, g["dssim_da"], rtol=2e-3, atol=3e-8)


@pytest.mark.parametrize("shape", [(3, 64, 96), (1, 5, 7), (3, 33, 31), (4, 100, 17), (3, 1, 1)])
def test_fused_loss_matches_oracle(shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.random(shape, dtype=np.float32)
    b = np.clip(a + 0.2 * rng.standard_normal(shape).astype(np.float32), 0, 1).astype(np.float32)
    b[..., : shape[2] // 2] = a[..., : shape[2] // 2]          # identical region: SSIM == 1, sign(0) == 0 there
    l1o, so, go = LO.loss_and_grad(a, b, 0.8, -0.2)
    l1, s, grad = _fused(a, b, 0.8, -0.2)
    assert abs(l1 - l1o) < 1e-6 and abs(s - so) < 2e-5
    scale = np.abs(go).max()
    assert np.abs(grad - go).max() <= 2e-4 * scale + 1e-9, np.abs(grad - go).max() / scale


def test_fused_loss_full_size_properties():
    """1080p: agrees with the separable PyTorch implementation, is bitwise reproducible, SSIM(x, x) == 1 with zero
    gradient, and the gradient is linear in the upstream scalars."""
    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 1080, 1920, generator=g).to(DEV)
    b = (a + 0.1 * torch.randn(3, 1080, 1920, generator=g).to(DEV)).clamp(0, 1)
    a1 = a.clone().requires_grad_(True); a2 = a.clone().requires_grad_(True)
    lf = L.fused_photometric_loss(a1, b); lf.backward()
    lt = L.photometric_loss(a2, b); lt.backward()
    assert abs(lf.item() - lt.item()) < 1e-5
    rel = (a1.grad - a2.grad).norm() / a2.grad.norm()
    assert rel.item() < 1e-4, rel.item()
    a3 = a.clone().requires_grad_(True)
    lf2 = L.fused_photometric_loss(a3, b); lf2.backward()
    assert torch.equal(lf, lf2) and torch.equal(a1.grad, a3.grad)
    a4 = a.clone().requires_grad_(True)
    l1, s = L.fused_l1_ssim(a4, a)
    assert l1.item() == 0.0 and abs(s.item() - 1.0) < 1e-6
    (3.0 * s).backward()
    assert a4.grad.abs().max().item() < 1e-9
    a5 = a.clone().requires_grad_(True)
    l1, s = L.fused_l1_ssim(a5, b); (2.0 * (0.8 * l1 - 0.2 * s)).backward()
    assert ((a5.grad - 2.0 * a1.grad).norm() / a1.grad.norm()).item() < 1e-6


@pytest.mark.parametrize("lam", [0.2, 0.0, 1.0, 0.37])
def test_combined_loss_equals_the_two_term_composition(lam):
    """bags_photometric_loss_* (the combination of train.py:325 inside the kernels) against the same expression written in
    PyTorch on the two terms: the loss to two ulp, dL/dimage to rounding, scaled upstream gradients,
    and the two logging terms detached."""
    g = torch.Generator().manual_seed(11)
    a = torch.rand(3, 70, 93, generator=g).to(DEV)
    b = (a + 0.1 * torch.randn(3, 70, 93, generator=g).to(DEV)).clamp(0, 1)
    a1 = a.clone().requires_grad_(True); a2 = a.clone().requires_grad_(True)
    loss, l1c, sc = L.fused_photometric_loss(a1, b, lam, return_terms=True)
    l1, s = L.fused_l1_ssim(a2, b)
    ref = (1.0 - lam) * l1 + lam * (1.0 - s)
    assert abs(loss.item() - ref.item()) <= 2.5e-7 and torch.equal(l1c, l1) and torch.equal(sc, s)     # (1 - lambda is rounded once on each side)
    assert not l1c.requires_grad and not sc.requires_grad and loss.requires_grad
    (2.5 * loss).backward(); (2.5 * ref).backward()
    scale = a2.grad.abs().max().item()
    assert (a1.grad - a2.grad).abs().max().item() <= 1e-6 * scale + 1e-12
    with pytest.raises(RuntimeError, match="lambda_dssim"):
        L.fused_photometric_loss(a1, b, 1.5)


def test_fused_loss_rejects_bad_arguments():
    a = torch.rand(3, 8, 8)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        L.fused_l1_ssim(a, a)
    with pytest.raises(RuntimeError, match="float32"):
        L.fused_l1_ssim(a.to(DEV).double(), a.to(DEV).double())
    with pytest.raises(RuntimeError, match="one shape"):
        L.fused_l1_ssim(a.to(DEV), torch.rand(3, 8, 9, device=DEV))
