"""View sharding across the GPUs of one node: one process per GPU, the Gaussian set replicated, rank r
renders views {v : v mod N == r} of the iteration's batch, accumulates their gradients locally, and the ONE exchange
step of the path is a sum over ranks of the Gaussian-parameter gradients (59 floats per Gaussian: xyz 3, f_dc 3,
f_rest 45, opacity 1, scale 3, rotation 4 -- scene/gaussian_model.py:197-204) over RCCL/xGMI.  Pose leaves belong to a
view, hence to one rank: never reduced.

The reference has no distributed code at all (it round-robins whole jobs over GPUs, high_resolution.sh:7-13); this is
new, so its contract is: N-rank summed gradients == 1-process sum over the same views (tests/test_sharding_cpu.py,
gloo world_size 2, incl. batches with fewer views than ranks).

Design for xGMI (point-to-point, 7 links per GPU; a ring collective is per-link bound, so what counts is few, large
collectives and as many views as possible behind each of them):
  * ``FlatGradBucket``: ONE persistent fp32 buffer per rank whose layout is derived from the parameter SHAPES only, so
    it is identical on every rank whatever a rank rendered.  Every ``p.grad`` is a view of it: autograd accumulates the
    gradients of all the rank's views in place, a rank without a view contributes its zeros, and the exchange is always
    the same single collective on the same length -- never one collective per tensor, never a rank-dependent layout.
  * V views per rank per exchange (the reference's cubemap step already renders 5 views per iteration,
    utils/cubemap_utils.py:229,263-265): local accumulation is free, the exchange is paid once per V views.
  * ``mode="all_reduce"``: one ``ncclAllReduce`` of the bucket.  ``mode="reduce_scatter"``: ``ncclReduceScatter`` ->
    optional per-shard hook (a sharded optimizer / clipping step sees only its 1/N of the rows) -> ``ncclAllGather``;
    same bytes on the links, and the hook's work is divided by N.
  * ``mode="sparse"`` (SURVEY.md 8e "scaling risk"): only the rows some rank touched cross the links.  A Gaussian no view of
    the batch sees has an all-zero gradient row on every rank, so leaving it out changes nothing: the ranks max-reduce a
    P-byte "row touched" mask, gather the touched rows of every parameter into one compact buffer, sum-all-reduce that
    (|U| x 59 floats instead of P x 59) and scatter the sums back.  Three collectives instead of one (the mask, the compact
    rows; the row count needs one host read), so it only pays when a batch sees a fraction of the scene: above
    ``dense_above`` (default 0.7 of the rows touched) the exchange falls back to the dense all-reduce.  The synthetic bench
    scenes are seen whole from every view and always fall back.
  * ``PipelinedExchange``: two buckets; the collective of batch k runs on RCCL's stream while batch k+1 renders into
    the other bucket (forward AND backward: nothing of batch k+1 depends on it unless the caller says so by waiting).
    Gradients arrive one batch late: a caller that steps its optimizer with them trades one step of staleness for an
    exchange that costs nothing on the critical path.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Round-robin view assignment: rank r gets r, r+N, r+2N, ..."""
    return list(range(rank, num_views, world))


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class FlatGradBucket:
    """One flat buffer for the gradients of ``params``; ``views[i]`` has the shape of ``params[i]``.

    The layout depends on the parameter shapes and ``world`` only (each slice starts on a 64-float boundary; the total
    is padded to a multiple of 64 * world so that a reduce-scatter splits it evenly) -- identical on every rank.  The
    padding is zeroed with the rest: the collective never reads uninitialised memory."""

    def __init__(self, params: Sequence[torch.Tensor], world: int = 1, align: int = 64):
        self.params = list(params)
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all bucketed parameters must share one device and dtype")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        unit = align * max(int(world), 1)
        self.numel = max((off + unit - 1) // unit * unit, unit)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]

    def zero_(self) -> None:
        self.flat.zero_()

    def bind(self) -> None:
        """Point every ``p.grad`` at its slice: autograd's accumulation (``p.grad += g``) then lands in the bucket."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def bound(self) -> bool:
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def absorb(self) -> None:
        """Safety net for callers that replaced a ``p.grad`` (``zero_grad(set_to_none=True)``, ``p.grad = None``) after
        ``bind``: whatever autograd left in a foreign tensor is added into the slice and the slice is bound again."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.add_(p.grad.reshape(v.shape))
                p.grad = v


class GradExchange:
    """Sum over ranks of one ``FlatGradBucket``, in place: always ONE collective (or one reduce-scatter + all-gather
    pair) of the same length on every rank."""

    def __init__(self, bucket: FlatGradBucket, group=None, mode: str = "all_reduce",
                 shard_hook: Optional[Callable[[torch.Tensor, int, int], None]] = None, dense_above: float = 0.7):
        if mode not in ("all_reduce", "reduce_scatter", "sparse"):
            raise ValueError("mode must be 'all_reduce', 'reduce_scatter' or 'sparse'")
        self.bucket, self.group, self.mode, self.shard_hook = bucket, group, mode, shard_hook
        self.dense_above = dense_above
        self.last_rows_exchanged: Optional[int] = None     # sparse mode: rows that crossed the links in the last exchange (P = dense fallback)
        if mode == "sparse":
            rows = {(p.shape[0] if p.dim() >= 1 else None) for p in bucket.params}
            if len(rows) != 1 or None in rows:
                raise ValueError("mode='sparse' needs parameters that all have one row per Gaussian (same first dimension)")
        self._pending: List[object] = []
        self._shard: Optional[torch.Tensor] = None
        self.collectives_issued = 0            # diagnostics / tests: collectives started so far

    def start(self) -> None:
        world = _world(self.group)
        if world == 1:
            if self.shard_hook is not None:
                self.shard_hook(self.bucket.flat, 0, self.bucket.numel)
            return
        flat = self.bucket.flat
        if self.mode == "sparse" and self._start_sparse():
            return
        if self.mode in ("all_reduce", "sparse"):
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.collectives_issued += 1
            return
        if flat.numel() % world:
            raise RuntimeError(f"bucket of {flat.numel()} elements was not laid out for world size {world}")
        n = flat.numel() // world
        rank = dist.get_rank(self.group)
        if self._shard is None or self._shard.numel() != n:
            self._shard = torch.empty(n, dtype=flat.dtype, device=flat.device)   # 1/N of the bucket, kept between steps
        shard = self._shard                     # out of place: not every backend accepts a shard aliasing its input
        w = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.collectives_issued += 1
        if self.shard_hook is not None:
            w.wait()                            # the hook reads the reduced shard (stream-ordered under RCCL)
            self.shard_hook(shard, rank * n, n)
        else:
            self._pending.append(w)
        self._pending.append(dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True))
        self.collectives_issued += 1

    def _start_sparse(self) -> bool:
        """Exchange only the rows touched on some rank.  Returns False when the dense all-reduce should run instead."""
        views = self.bucket.views
        P = views[0].shape[0]
        if P == 0:
            return True                                       # nothing to exchange (every rank holds the same empty set)
        touched = torch.zeros(P, dtype=torch.uint8, device=views[0].device)
        for v in views:                                       # a row is touched if any parameter's gradient row is non-zero
            touched |= (v.reshape(P, -1) != 0).any(dim=1).to(torch.uint8)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=self.group)
        self.collectives_issued += 1
        idx = touched.nonzero(as_tuple=False).squeeze(1)      # the one host read: the compact collective's length
        n = int(idx.numel())
        if n > self.dense_above * P:
            self.last_rows_exchanged = P
            return False
        self.last_rows_exchanged = n
        if n == 0:
            return True                                       # nobody saw anything: every row is zero everywhere
        widths = [v[0].numel() for v in views]
        compact = torch.empty(n, sum(widths), dtype=views[0].dtype, device=views[0].device)
        off = 0
        for v, w in zip(views, widths):
            compact[:, off:off + w] = v.reshape(P, w).index_select(0, idx)
            off += w
        dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=self.group)
        self.collectives_issued += 1
        off = 0
        for v, w in zip(views, widths):
            v.reshape(P, w).index_copy_(0, idx, compact[:, off:off + w])
            off += w
        return True

    def wait(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending = []

    def run(self) -> None:
        self.start()
        self.wait()


class GradAllReducer:
    """Sum-all-reduce of ``.grad`` of the replicated Gaussian parameters through a persistent ``FlatGradBucket``.

    Usage per iteration: ``begin()`` (zero the bucket, bind ``p.grad``), any number of ``loss.backward()`` calls, then
    ``all_reduce()`` (or ``start()`` ... ``wait()``).  ``all_reduce()`` without a preceding ``begin()`` still works:
    gradients autograd put elsewhere are absorbed into the bucket first (one extra copy).

    ``all_reduce_single_view()`` is the ONE-view-per-rank-per-exchange fast path (BASELINE configs 3-5): the rasterizer's
    backward already carves its Gaussian gradients out of one buffer, autograd hands those tensors over as ``p.grad`` without
    a copy, and the collective runs on that buffer as it is -- no bucket to zero (25 us at 500 k Gaussians), no accumulation
    pass (60 us).  The caller guarantees that EVERY rank differentiated exactly one view through the op with ``p.grad``
    cleared beforehand (the layout is then identical on all ranks); anything else raises instead of hanging a collective."""

    def __init__(self, params: Sequence[torch.Tensor], group=None, mode: str = "all_reduce", shard_hook=None,
                 dense_above: float = 0.7):
        self.params = list(params)
        self.group = group
        self.bucket = FlatGradBucket(self.params, _world(group))
        self.exchange = GradExchange(self.bucket, group, mode, shard_hook, dense_above)
        self.single_view_collectives = 0

    def all_reduce_single_view(self) -> None:
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("all_reduce_single_view: a parameter has no gradient (every rank must have rendered one view)")
        st = grads[0].untyped_storage()
        esz = grads[0].element_size()
        for g in grads:
            if g.untyped_storage().data_ptr() != st.data_ptr() or not g.is_contiguous() or g.dtype != grads[0].dtype:
                raise RuntimeError("all_reduce_single_view: the gradients are not views of one buffer (they did not come straight "
                                   "from one rasterizer backward); use begin() ... all_reduce() instead")
        # The collective's length and the position of every slice must be the same on all ranks.  The op's backward carves
        # its Gaussian gradients in a fixed order on 64-float boundaries (rasterizer.py, `want`), so the layout is a function
        # of the parameter shapes alone -- PROVIDED nothing else sits in the buffer: the slices, taken in buffer order, must
        # pack exactly like a FlatGradBucket of those shapes.  A rank whose gradients do not (another set of inputs requiring
        # gradients, tensors from two different backwards) raises here instead of corrupting or hanging the collective.
        order = sorted(range(len(grads)), key=lambda i: grads[i].storage_offset())
        base = grads[order[0]].storage_offset()
        off, gaps = 0, []
        for i in order:
            g = grads[i]
            if g.storage_offset() - base != off or g.numel() != self.params[i].numel():
                raise RuntimeError("all_reduce_single_view: the gradient buffer is not one 64-float aligned packing of the "
                                   "parameters' gradients; use begin() ... all_reduce() instead")
            end = off + g.numel()
            off = (end + 63) // 64 * 64
            if i != order[-1]:
                gaps.append((end, off))
        if base + end > st.nbytes() // esz:
            raise RuntimeError("all_reduce_single_view: gradient buffer shorter than its layout")
        if _world(self.group) == 1:
            return
        flat = torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(st, base, (end,))
        # the alignment gaps between the slices come from torch.empty(): zeroed here (one indexed fill, cached index) so that
        # the collective never reads uninitialised memory -- on this path only, where a 0.6 ms collective follows
        key = (flat.device, tuple(gaps))
        if getattr(self, "_gaps", (None, None))[0] != key:
            idx = [j for a_, b_ in gaps for j in range(a_, b_)]
            self._gaps = (key, torch.tensor(idx, dtype=torch.int64, device=flat.device) if idx else None)
        if self._gaps[1] is not None:
            flat.index_fill_(0, self._gaps[1], 0.0)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        self.single_view_collectives += 1

    def all_reduce_adopted(self) -> None:
        """The same collective after SEVERAL views of this rank: with every ``p.grad`` None before the first view, the first
        backward's carved buffer becomes ``p.grad`` (handed over without a copy) and autograd adds the later views into it in
        place -- the buffer is the accumulator, no bucket zero pass and no add for the first view."""
        self.all_reduce_single_view()

    def begin(self) -> None:
        self.bucket.zero_()
        self.bucket.bind()

    def start(self) -> None:
        if not self.bucket.bound():
            self.bucket.absorb()
        self.exchange.start()

    def wait(self) -> None:
        self.exchange.wait()

    def all_reduce(self) -> None:
        self.start()
        self.wait()


class PipelinedExchange:
    """Two buckets: while the collective of batch k is in flight on the communication stream, batch k+1 accumulates
    into the other bucket.  ``begin()`` returns nothing and binds the free bucket; ``submit()`` starts the collective of
    the bucket just filled; ``reduced()`` waits for and returns the views of the OLDEST submitted bucket (one batch
    late).  ``drain()`` waits for everything (end of an epoch / before anything reads ``p.grad`` directly)."""

    def __init__(self, params: Sequence[torch.Tensor], group=None, mode: str = "all_reduce"):
        world = _world(group)
        self.buckets = [FlatGradBucket(params, world), FlatGradBucket(params, world)]
        self.exchanges = [GradExchange(b, group, mode) for b in self.buckets]
        self._cur = 0
        self._in_flight: List[int] = []

    def begin(self) -> None:
        if self._cur in self._in_flight:          # its collective must have finished before it is zeroed again
            self.exchanges[self._cur].wait()
            self._in_flight.remove(self._cur)
        self.buckets[self._cur].zero_()
        self.buckets[self._cur].bind()

    def submit(self) -> None:
        b = self.buckets[self._cur]
        if not b.bound():
            b.absorb()
        self.exchanges[self._cur].start()
        self._in_flight.append(self._cur)
        self._cur ^= 1

    def reduced(self) -> Optional[List[torch.Tensor]]:
        if not self._in_flight:
            return None
        i = self._in_flight.pop(0)
        self.exchanges[i].wait()
        return self.buckets[i].views

    def drain(self) -> None:
        for i in self._in_flight:
            self.exchanges[i].wait()
        self._in_flight = []


class ViewShardedRenderer:
    """Renders this rank's share of a batch of views and leaves summed gradients on every rank.

    ``render_fn(view) -> scalar loss`` must run forward for one view and return the loss whose backward populates the
    shared parameters' ``.grad`` (the product passes a closure over bags_raster.GaussianRasterizer; the gloo CPU tests
    pass a closure over the oracle -- the sharding logic is identical).  A batch of N*V views is V views per rank behind
    ONE exchange; batches that do not divide evenly, or hold fewer views than ranks, are fine: every rank always joins
    the same collective on the same bucket."""

    def __init__(self, params: Sequence[torch.Tensor], render_fn: Callable[[object], torch.Tensor], group=None,
                 mode: str = "all_reduce", sh_params: Optional[Sequence[torch.Tensor]] = None):
        """sh_params (optional): ``(means3D, shs)`` or ``(means3D, features_dc, features_rest)`` -- the leaves the rasterizer is called
        with.  With them the SH gradients of the rank's views stay factored (rasterizer.FactoredSH: 12 bytes per Gaussian and view) and
        one pass forms the step's rows before the exchange -- the same sums bit for bit, a third less traffic per view from the second
        view of a rank on."""
        self.params = list(params)
        self.render_fn = render_fn
        self.group = group
        self.reducer = GradAllReducer(self.params, group, mode)
        self.sh_params = tuple(sh_params) if sh_params is not None else None
        if self.sh_params is not None and len(self.sh_params) not in (2, 3):
            raise ValueError("sh_params must be (means3D, shs) or (means3D, features_dc, features_rest)")

    def step(self, views: Sequence[object]) -> Dict[str, object]:
        world = _world(self.group)
        rank = dist.get_rank(self.group) if world > 1 else 0
        self.reducer.begin()
        mine = shard_views(len(views), rank, world)
        losses = []
        # the parameters' .grad are the (zeroed) bucket slices: where the rasterizer is called on the leaves themselves its backward
        # adds into them inside the kernel (rasterizer.ACCUMULATE_IN_PLACE) instead of through one autograd add pass per tensor and view
        from . import rasterizer as _R
        saved = (_R.ACCUMULATE_IN_PLACE, _R.FACTORED_SH)
        _R.ACCUMULATE_IN_PLACE = True
        fs = _R.FactoredSH() if (self.sh_params is not None and self.sh_params[0].is_cuda) else None
        _R.FACTORED_SH = fs
        try:
            for v in mine:
                loss = self.render_fn(views[v])
                loss.backward()                  # grads of this rank's views accumulate in the bucket
                losses.append(loss.detach())
        finally:
            _R.ACCUMULATE_IN_PLACE, _R.FACTORED_SH = saved
        if fs is not None:
            fs.finish(*self.sh_params)           # the step's SH-gradient rows, added into the bucket's (zeroed) slices
        self.reducer.all_reduce()
        total = torch.stack(losses).sum() if losses else torch.zeros((), device=self.params[0].device)
        if world > 1:
            dist.all_reduce(total, group=self.group)
        return {"loss_sum": total, "views": mine}


class DensificationSync:
    """Keeps the densification statistics of a view-sharded run equal to the single-process ones.

    Every rank feeds ``GaussianBag.add_densification_stats`` with ITS views only, so ``xyz_gradient_accum`` / ``denom``
    (sums over views, scene/gaussian_model.py:449-455) and ``max_radii2D`` (maximum over views, train.py:377,400) drift
    apart between ranks.  ``sync`` exchanges what each rank added since the previous call: sum all-reduce of the
    increments, max all-reduce of the radii.  Call it before anything that reads the statistics (densify_and_prune), on
    every rank.  A no-op without a process group.  After the statistics were reset, re-sized or RESTORED from a
    checkpoint (``bags_raster.io.load_checkpoint`` + ``restore``), call ``rebase()`` before the next ``sync``."""

    def __init__(self, pc, group=None):
        self.pc, self.group = pc, group
        self._base_accum = pc.xyz_gradient_accum.detach().clone()
        self._base_denom = pc.denom.detach().clone()

    def sync(self) -> None:
        pc = self.pc
        if _world(self.group) > 1:
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
        """After the statistics were reset or re-sized (densification_postfix zeroes them, scene/gaussian_model.py:388-391)
        or restored from a checkpoint."""
        self._base_accum = self.pc.xyz_gradient_accum.detach().clone()
        self._base_denom = self.pc.denom.detach().clone()
