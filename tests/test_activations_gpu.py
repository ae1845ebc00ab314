"""Fused parameter activations (csrc/activations.hip behind bags_activations_forward / _backward) against the PyTorch
properties of GaussianBag, which tests/test_golden_cpu.py pins to the reference (scene/gaussian_model.py:118-141)."""
import pytest
import torch

from bags_raster.gaussians import GaussianBag, fused_activations
from bags_raster.synth import synth_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("P,deg", [(1000, 3), (257, 0), (1, 2)])
def test_fused_activations_match_properties(P, deg):
    sc = synth_scene(P, 9, 0.5, deg)
    g = torch.Generator().manual_seed(2)
    sc["rotations"] = sc["rotations"] * (0.3 + 2.0 * torch.rand(P, 1, generator=g))        # un-normalised quaternions
    pc0 = GaussianBag.from_activated(sc, deg, device=DEV)
    pc1 = GaussianBag.from_activated(sc, deg, device=DEV)
    with torch.no_grad():
        for a, b in zip(pc0.leaves(), pc1.leaves()):
            b.copy_(a)
        pc0._rotation.copy_(sc["rotations"].to(DEV)); pc1._rotation.copy_(sc["rotations"].to(DEV))
    want = [pc0.get_xyz, pc0.get_features, pc0.get_opacity, pc0.get_scaling, pc0.get_rotation]
    got = list(pc1.activated())
    cots = [torch.randn(w.shape, generator=g).to(DEV) for w in want]
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-6, atol=1e-7), (a - b).abs().max()
    torch.autograd.backward(want[1:], cots[1:])
    torch.autograd.backward(got[1:], cots[1:])
    for a, b in zip(pc1.leaves()[1:], pc0.leaves()[1:]):
        if a.numel() == 0:                                  # degree 0: features_rest is (P,0,3)
            continue
        assert a.grad is not None and torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-5), (a.grad - b.grad).abs().max()   # (g - q (q.g)) / |q| cancels


def test_fused_activations_partial_use_and_errors():
    sc = synth_scene(300, 1, 0.5, 3)
    pc = GaussianBag.from_activated(sc, 3, device=DEV)
    _, shs, op, scl, rot = pc.activated()
    (scl.sum() + op.sum()).backward()                     # shs and rotations unused: their leaves get no gradient
    assert pc._features_dc.grad is None and pc._features_rest.grad is None and pc._rotation.grad is None
    assert torch.allclose(pc._scaling.grad, torch.exp(pc._scaling.detach()))
    with pytest.raises(RuntimeError, match="GPU"):
        fused_activations(torch.zeros(2, 1, 3), torch.zeros(2, 15, 3), torch.zeros(2, 1), torch.zeros(2, 3), torch.ones(2, 4))



def test_activations_without_the_feature_concatenation():
    """activated(features=False): the three activations from one launch with one thread per Gaussian, no (P,K,3) tensor."""
    sc = synth_scene(500, 4, 0.5, 3)
    pc0 = GaussianBag.from_activated(sc, 3, device=DEV)
    pc1 = GaussianBag.from_activated(sc, 3, device=DEV)
    want, got = list(pc0.activated()), list(pc1.activated(features=False))
    assert got[1] is None
    g = torch.Generator().manual_seed(6)
    for j in (2, 3, 4):
        assert torch.equal(got[j], want[j])
    cots = [torch.randn(w.shape, generator=g).to(DEV) for w in want]
    torch.autograd.backward(want[2:], cots[2:]); torch.autograd.backward(got[2:], cots[2:])
    for a, b in zip(pc1.leaves()[3:], pc0.leaves()[3:]):
        assert torch.equal(a.grad, b.grad)
    assert pc1._features_dc.grad is None and pc1._features_rest.grad is None
