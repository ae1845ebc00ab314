"""CPU restatement of distCUDA2 -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

PARITY UNPINNED: simple-knn is an empty submodule of the reference (.gitmodules:1-3, no commit pin, no tests), so there
is no reference output to check against.  The contract restated here is the published one [UPSTREAM-KNOWLEDGE], anchored
on the call site scene/gaussian_model.py:177-178 (`dist2 = clamp_min(distCUDA2(points), 1e-7)`; scales = log sqrt):
for every point the mean of the squared Euclidean distances to its three nearest neighbours, the point itself excluded
by index (coincident points count with distance 0), squared distance formed as dx*dx + dy*dy + dz*dz in float32, missing
neighbours (fewer than four points) contributing FLT_MAX.  Brute force, chunked; tests cross-check it against
scipy.spatial.cKDTree, an independent exact k-NN."""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def dist_mean3(points: np.ndarray, chunk: int = 1024) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        q = p[s:s + chunk]
        d = q[:, None, :] - p[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]          # float32
        d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf                    # self, by index
        k = min(3, n - 1)
        best = np.full((q.shape[0], 3), FLT_MAX, dtype=np.float32)
        if k > 0:
            best[:, :k] = np.sort(np.partition(d2, k - 1, axis=1)[:, :k], axis=1)
        with np.errstate(over="ignore"):
            out[s:s + chunk] = (best[:, 0] + best[:, 1] + best[:, 2]) / np.float32(3.0)
    return out
