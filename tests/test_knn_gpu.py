"""distCUDA2 (csrc/knn.hip behind bags_knn_mean_dist2) against the brute-force oracle and scipy's cKDTree."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

from bags_raster.knn import distCUDA2
from oracle import knn_oracle as KO

pytestmark = pytest.mark.gpu


def _hip(pts: np.ndarray) -> np.ndarray:
    return distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()


def _cloud(kind, n, rng):
    if kind == "uniform":
        return (rng.random((n, 3)) * 2.6 - 1.3).astype(np.float32)
    if kind == "clustered":                      # dense blobs + a few far outliers: most grid cells are empty
        c = rng.normal(size=(8, 3)) * 5.0
        p = c[rng.integers(0, 8, n)] + rng.normal(size=(n, 3)) * 0.05
        p[:5] = rng.normal(size=(5, 3)) * 400.0
        return p.astype(np.float32)
    if kind == "plane":                          # degenerate extent along z
        p = rng.random((n, 3)); p[:, 2] = 0.25
        return p.astype(np.float32)
    if kind == "line":
        p = np.zeros((n, 3)); p[:, 0] = rng.random(n) * 100.0
        return p.astype(np.float32)
    if kind == "duplicates":
        p = rng.random((n // 4, 3))
        return np.repeat(p, 4, axis=0).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "plane", "line", "duplicates"])
def test_knn_matches_oracle(kind):
    rng = np.random.default_rng(11)
    pts = _cloud(kind, 4000, rng)
    got, want = _hip(pts), KO.dist_mean3(pts)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 257])
def test_knn_tiny_inputs(n):
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3)).astype(np.float32)
    got, want = _hip(pts), KO.dist_mean3(pts)
    big = want > 1e37                            # fewer than three neighbours: FLT_MAX terms (sum overflows to inf)
    assert np.array_equal(got > 1e37, big)
    np.testing.assert_allclose(got[~big], want[~big], rtol=2e-6)


def test_knn_full_size_against_kdtree_and_call_site():
    """500 k points of the bench scene: exact agreement with cKDTree, and the reference's use of the result
    (scene/gaussian_model.py:177-178) gives finite log-scales."""
    from bags_raster.synth import synth_scene
    pts = synth_scene(500_000, 0, 0.5, 0)["means3D"].numpy()
    t = torch.from_numpy(pts).cuda()
    distCUDA2(t); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = distCUDA2(t); e1.record(); torch.cuda.synchronize()
    print("distCUDA2(500k): %.3f ms" % e0.elapsed_time(e1))
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
    want = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-5)
    scales = torch.log(torch.sqrt(torch.clamp_min(out, 1e-7)))[..., None].repeat(1, 3)
    assert torch.isfinite(scales).all()
    # the alias the reference imports
    from simple_knn._C import distCUDA2 as alias
    assert torch.equal(alias(t), out)


def test_knn_rejects_cpu_and_bad_shapes():
    with pytest.raises(RuntimeError, match="GPU tensor"):
        distCUDA2(torch.rand(10, 3))
    with pytest.raises(RuntimeError, match=r"\(P,3\)"):
        distCUDA2(torch.rand(10, 2).cuda())
