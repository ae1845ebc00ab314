"""distCUDA2 -- mean squared distance to the three nearest neighbours, the scale initialiser of ``create_from_pcd``
(scene/gaussian_model.py:177-178: ``dist2 = clamp_min(distCUDA2(points), 1e-7); scales = log(sqrt(dist2))``).

Mirror of ``simple_knn._C.distCUDA2`` (the reference's second native dependency, scene/gaussian_model.py:20) on top of
``bags_knn_mean_dist2`` (include/bags_raster.h, csrc/knn.hip).  GPU tensors only; there is no fallback."""
from __future__ import annotations

import torch

from . import _lib as L


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be a GPU tensor (no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError(f"distCUDA2: expected (P,3) points, got {tuple(points.shape)}")
    pts = points.detach().to(torch.float32).contiguous()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    lib = L.load()
    with torch.cuda.device(pts.device):
        nbytes = lib.bags_knn_workspace_size(P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
        L.check(lib.bags_knn_mean_dist2(pts.data_ptr(), P, ws.data_ptr(), nbytes, out.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "bags_knn_mean_dist2")
    return out
