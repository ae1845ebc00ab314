"""View sharding across the GPUs of one node: one process per GPU, the Gaussian set replicated, rank r
renders views {v : v mod N == r} of the iteration's batch, and the ONE exchange step of the path is a sum all-reduce of
the Gaussian-parameter gradients (59 floats per Gaussian: xyz 3, f_dc 3, f_rest 45, opacity 1, scale 3, rotation 4 --
scene/gaussian_model.py:197-204) over RCCL/xGMI.  Pose leaves belong to a view, hence to one rank: never reduced.

The reference has no distributed code at all (it round-robins whole jobs over GPUs, high_resolution.sh:7-13); this is
new, so its contract is: N-rank summed gradients == 1-process sum over the same views (tests/test_sharding_cpu.py,
gloo world_size 2).

xGMI is point-to-point (7 links/GPU): a 118 MB (P = 500 k) ring all-reduce is link-bound, so the reducer issues the
per-tensor collectives asynchronously, largest first, on RCCL's own stream while the caller may keep enqueueing the
next view's forward; `wait()` joins them.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Round-robin view assignment: rank r gets r, r+N, r+2N, ..."""
    return list(range(rank, num_views, world))


def coalesce_by_storage(grads: Sequence[torch.Tensor], max_waste: float = 0.02) -> List[torch.Tensor]:
    """Tensors to all-reduce in place of ``grads``: contiguous tensors of one dtype that were carved out of one buffer
    (bags_raster.rasterizer's backward does that for the Gaussian-parameter gradients) are replaced by ONE flat view
    spanning them, provided the padding between them stays below ``max_waste`` of the span and the storage holds nothing
    else inside it that the sum could disturb (the span only covers bytes between the first and last tensor; padding is
    uninitialised but never read).  Everything else is passed through.  One large collective instead of five: the
    per-call latency of a ring over 8 GPUs is paid once."""
    groups: Dict[tuple, List[torch.Tensor]] = {}
    out: List[torch.Tensor] = []
    for g in grads:
        if g.is_contiguous() and g.layout == torch.strided and g.numel() > 0:
            groups.setdefault((g.untyped_storage().data_ptr(), g.dtype, g.device), []).append(g)
        else:
            out.append(g)
    for (_, dtype, device), ts in groups.items():
        if len(ts) == 1:
            out.append(ts[0])
            continue
        ts = sorted(ts, key=lambda t: t.storage_offset())
        lo, hi = ts[0].storage_offset(), max(t.storage_offset() + t.numel() for t in ts)
        used = sum(t.numel() for t in ts)
        overlap = any(a.storage_offset() + a.numel() > b.storage_offset() for a, b in zip(ts, ts[1:]))
        if overlap or used < (1.0 - max_waste) * (hi - lo):
            out.extend(ts)
            continue
        flat = torch.empty(0, dtype=dtype, device=device).set_(ts[0].untyped_storage(), lo, (hi - lo,), (1,))
        out.append(flat)
    return out


class GradAllReducer:
    """Sum-all-reduce of ``.grad`` of the replicated Gaussian parameters, in place, asynchronously."""

    def __init__(self, params: Sequence[torch.Tensor], group=None):
        self.params = list(params)
        self.group = group
        self._pending = []

    def start(self) -> None:
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        for g in sorted(coalesce_by_storage(grads), key=lambda t: -t.numel()):
            self._pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending = []

    def all_reduce(self) -> None:
        self.start()
        self.wait()


class ViewShardedRenderer:
    """Renders this rank's share of a batch of views and leaves summed gradients on every rank.

    ``render_fn(view) -> scalar loss`` must run forward for one view and return the loss whose backward populates the
    shared parameters' ``.grad`` (the product passes a closure over bags_raster.GaussianRasterizer; the gloo CPU tests
    pass a closure over the oracle -- the sharding logic is identical)."""

    def __init__(self, params: Sequence[torch.Tensor], render_fn: Callable[[object], torch.Tensor], group=None):
        self.params = list(params)
        self.render_fn = render_fn
        self.group = group
        self.reducer = GradAllReducer(self.params, group)

    def step(self, views: Sequence[object]) -> Dict[str, object]:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        for p in self.params:
            p.grad = None
        mine = shard_views(len(views), rank, world)
        losses = []
        for v in mine:
            loss = self.render_fn(views[v])
            loss.backward()                      # grads of this rank's views accumulate locally
            losses.append(loss.detach())
        for p in self.params:                    # a rank with no view still joins the collective
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self.reducer.all_reduce()
        total = torch.stack(losses).sum() if losses else torch.zeros((), device=self.params[0].device)
        if world > 1:
            dist.all_reduce(total, group=self.group)
        return {"loss_sum": total, "views": mine}


class DensificationSync:
    """Keeps the densification statistics of a view-sharded run equal to the single-process ones.

    Every rank feeds ``GaussianBag.add_densification_stats`` with ITS views only, so ``xyz_gradient_accum`` / ``denom``
    (sums over views, scene/gaussian_model.py:449-455) and ``max_radii2D`` (maximum over views, train.py:377,400) drift apart between ranks.  ``sync`` exchanges what each rank
    added since the previous call: sum all-reduce of the increments, max all-reduce of the radii.  Call it before anything
    that reads the statistics (densify_and_prune), on every rank.  A no-op without a process group."""

    def __init__(self, pc, group=None):
        self.pc, self.group = pc, group
        self._base_accum = pc.xyz_gradient_accum.detach().clone()
        self._base_denom = pc.denom.detach().clone()

    def sync(self) -> None:
        pc = self.pc
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            d_accum = pc.xyz_gradient_accum.detach() - self._base_accum
            d_denom = pc.denom.detach() - self._base_denom
            work = [dist.all_reduce(d_accum, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                    dist.all_reduce(d_denom, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                    dist.all_reduce(pc.max_radii2D, op=dist.ReduceOp.MAX, group=self.group, async_op=True)]
            for w in work:
                w.wait()
            pc.xyz_gradient_accum = self._base_accum + d_accum
            pc.denom = self._base_denom + d_denom
        self._base_accum = pc.xyz_gradient_accum.detach().clone()
        self._base_denom = pc.denom.detach().clone()

    def rebase(self) -> None:
        """After the statistics were reset or re-sized (densification_postfix zeroes them, scene/gaussian_model.py:388-391)."""
        self._base_accum = self.pc.xyz_gradient_accum.detach().clone()
        self._base_denom = self.pc.denom.detach().clone()

