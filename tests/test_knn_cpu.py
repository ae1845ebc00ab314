"""oracle/knn_oracle.py against an independent exact k-NN (scipy cKDTree)."""
import numpy as np
from scipy.spatial import cKDTree

from oracle import knn_oracle as KO


def test_knn_oracle_matches_kdtree():
    rng = np.random.default_rng(0)
    pts = (rng.random((3000, 3)) * 2.6 - 1.3).astype(np.float32)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(KO.dist_mean3(pts), want, rtol=2e-5)


def test_knn_oracle_edge_cases():
    one = KO.dist_mean3(np.zeros((1, 3), np.float32))
    assert one[0] >= 3e38                                            # no neighbours: FLT_MAX contributions
    two = KO.dist_mean3(np.array([[0, 0, 0], [1, 0, 0]], np.float32))
    assert np.all(two >= 1e38)
    dup = KO.dist_mean3(np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0], [2, 0, 0]], np.float32))
    np.testing.assert_allclose(dup, [0, 0, 0, 0, 4.0], atol=0)       # coincident points count with distance 0
