"""World-size-2 gloo tests of the view-sharded path: N-rank summed gradients == 1-process sum over the
same views.  The render inside each rank is the CPU oracle (the HIP op needs a GPU); the sharding / all-reduce logic
under test is the product's (bags_raster.sharding), identical for the RCCL backend."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _views_and_scene(n=5):
    from bags_raster.synth import sphere_views, synth_scene
    scene = synth_scene(320, 3, 3.0, 1)                    # 320: every gradient is a multiple of 64 floats (no padding in the flat buffer)
    cams = sphere_views(n, 48, 32, noise=0.05)            # 5 views: uneven split over 2 ranks (3 + 2)
    return scene, cams


def _make_render_fn(params):
    from oracle import raster_oracle as O
    from scenes import oracle_settings

    class _Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, cam, *tensors):
            names = ("means3D", "scales", "rotations", "opacities", "shs")
            inp = dict(zip(names, tensors)); inp["shift_factors"] = torch.zeros(3)
            s = oracle_settings(cam, 1)
            ctx.inp, ctx.s = inp, s
            st, _ = O.render_and_grad(inp, s, None)
            return st.image

        @staticmethod
        def backward(ctx, g):
            from oracle import raster_oracle as O2
            with torch.enable_grad():
                _, gr = O2.render_and_grad(ctx.inp, ctx.s, g)
            # like the product's backward (bags_raster/rasterizer.py): the gradients are views of ONE buffer, 64-float
            # aligned, so that the exchange is a single collective
            names = ("means3D", "scales", "rotations", "opacities", "shs")
            sizes = [gr[n].numel() for n in names]
            flat = torch.empty(sum((v + 63) // 64 * 64 for v in sizes), dtype=gr["means3D"].dtype)
            views, off = [], 0
            for n, v in zip(names, sizes):
                views.append(flat[off:off + v].view(gr[n].shape).copy_(gr[n]))
                off += (v + 63) // 64 * 64
            del flat
            return (None,) + tuple(views)

    def render(cam):
        img = _Op.apply(cam, *params)
        target = torch.full_like(img, 0.25)
        return ((img - target) ** 2).mean()
    return render


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _params(scene):
    return [scene[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")]


def _worker(rank, world, port, out, n_views, mode, iters=2):
    _setup(rank, world, port)
    from bags_raster.sharding import ViewShardedRenderer, shard_views
    scene, cams = _views_and_scene(max(5, n_views))
    cams = cams[:n_views]
    params = _params(scene)
    r = ViewShardedRenderer(params, _make_render_fn(params), mode=mode)
    for it in range(iters):                               # twice: the bucket must be zeroed between iterations
        res = r.step(cams)
    assert res["views"] == shard_views(len(cams), rank, world)
    # ONE exchange per step whatever this rank rendered (no view at all included): 1 all-reduce, or 1 reduce-scatter + 1 all-gather
    assert r.reducer.exchange.collectives_issued == iters * (1 if mode == "all_reduce" else 2)      # sparse: mask + rows (or dense)
    assert r.reducer.bucket.bound(), "p.grad is no longer a view of the flat bucket"
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in params], "loss": res["loss_sum"]}, out)
    for p in params:                                      # every rank must hold the same summed gradients
        ref = p.grad.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, p.grad)
    dist.destroy_process_group()


def test_shard_views_round_robin():
    from bags_raster.sharding import shard_views
    assert shard_views(5, 0, 2) == [0, 2, 4] and shard_views(5, 1, 2) == [1, 3]
    assert shard_views(3, 5, 8) == [] and sorted(sum((shard_views(200, r, 8) for r in range(8)), [])) == list(range(200))


def _single_process_reference(n_views):
    scene, cams = _views_and_scene(max(5, n_views))
    params = _params(scene)
    render = _make_render_fn(params)
    total = 0.0
    for c in cams[:n_views]:
        loss = render(c)
        loss.backward()
        total += float(loss)
    return params, total


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_views,mode", [(5, "all_reduce"),        # V = 3 and 2 views behind one exchange (uneven split)
                                          (5, "reduce_scatter"),    # the same through reduce-scatter + all-gather
                                          (5, "sparse"),            # only the rows some rank touched (here: all -> dense fallback)
                                          (1, "all_reduce")])       # fewer views than ranks: rank 1 renders nothing
def test_two_rank_sum_equals_single_process(tmp_path, n_views, mode):
    out = str(tmp_path / "rank0.pt")
    port = 29500 + ((os.getpid() + 7 * n_views + len(mode)) % 500)
    mp.spawn(_worker, args=(2, port, out, n_views, mode), nprocs=2, join=True)
    got = torch.load(out)
    params, total = _single_process_reference(n_views)
    assert abs(float(got["loss"]) - total) < 1e-5 * max(1.0, abs(total))
    for g2, p in zip(got["grads"], params):
        denom = p.grad.norm().item()
        assert (g2 - p.grad).norm().item() <= 1e-6 * max(denom, 1e-12), "N-rank sum differs from 1-process sum"


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_views,mode,iters", [(8, "all_reduce", 2),        # BASELINE config 5's split: 8 views per iteration, one per rank
                                                (5, "reduce_scatter", 2),    # a batch smaller than the node: ranks 5, 6, 7 render nothing
                                                (200, "all_reduce", 1)])     # BASELINE config 4's split: 200 views, 25 per rank behind one exchange
def test_eight_rank_sum_equals_single_process(tmp_path, n_views, mode, iters):
    """The partitioning the 8-GPU runs will execute first (views v = r mod 8, Gaussians replicated, ONE exchange
    per step on a buffer whose layout does not depend on what a rank rendered) at world size 8 -- rounds 1-4 only ever ran it at
    world size 2.  gloo on the CPU; the collective calls are the ones the RCCL backend gets."""
    out = str(tmp_path / "rank0.pt")
    port = 29500 + ((os.getpid() + 11 * n_views + len(mode)) % 500)
    mp.spawn(_worker, args=(8, port, out, n_views, mode, iters), nprocs=8, join=True)
    got = torch.load(out)
    params, total = _single_process_reference(n_views)
    assert abs(float(got["loss"]) - total) < 1e-5 * max(1.0, abs(total))
    for g2, p in zip(got["grads"], params):
        denom = p.grad.norm().item()
        assert (g2 - p.grad).norm().item() <= 2e-6 * max(denom, 1e-12), "8-rank sum differs from 1-process sum"


def _raw_leaf_worker(rank, world, port, out):
    """The training path: raw (pre-activation) leaves of a GaussianBag -> activations -> op.  The leaf gradients come out
    of the activation backward as unrelated tensors; the exchange must still be ONE collective."""
    _setup(rank, world, port)
    from bags_raster.gaussians import GaussianBag
    from bags_raster.sharding import ViewShardedRenderer
    scene, cams = _views_and_scene()
    pc = GaussianBag.from_activated(scene, 1)
    leaves = list(pc.leaves())
    acts = lambda: [pc.get_xyz, pc.get_scaling, pc.get_rotation, pc.get_opacity, pc.get_features]

    class _Lazy(list):                                     # _make_render_fn reads `params` at call time
        def __iter__(self):
            return iter(acts())
    render = _make_render_fn(_Lazy())
    r = ViewShardedRenderer(leaves, render)
    r.step(cams)
    assert r.reducer.exchange.collectives_issued == 1
    assert all(p.grad is not None and p.grad.abs().sum() > 0 for p in leaves)
    if rank == 0:
        torch.save([p.grad.clone() for p in leaves], out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_raw_leaves_of_a_gaussian_bag_exchange_as_one_collective(tmp_path):
    out = str(tmp_path / "leaves.pt")
    port = 29500 + ((os.getpid() + 191) % 500)
    mp.spawn(_raw_leaf_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    from bags_raster.gaussians import GaussianBag
    scene, cams = _views_and_scene()
    pc = GaussianBag.from_activated(scene, 1)

    class _Lazy(list):
        def __iter__(self):
            return iter([pc.get_xyz, pc.get_scaling, pc.get_rotation, pc.get_opacity, pc.get_features])
    render = _make_render_fn(_Lazy())
    for c in cams:
        render(c).backward()
    for g2, p in zip(got, pc.leaves()):
        assert (g2 - p.grad).norm().item() <= 1e-6 * max(p.grad.norm().item(), 1e-12)


def _pipelined_worker(rank, world, port, out):
    _setup(rank, world, port)
    from bags_raster.sharding import PipelinedExchange, shard_views
    scene, cams = _views_and_scene()
    params = _params(scene)
    render = _make_render_fn(params)
    pipe = PipelinedExchange(params)
    batches = [cams[0:2], cams[2:5], cams[1:4]]           # three batches: both buckets get reused
    got = []
    for batch in batches:
        pipe.begin()
        for v in shard_views(len(batch), rank, world):
            render(batch[v]).backward()
        pipe.submit()
        if len(pipe._in_flight) == 2:                     # gradients of the batch before this one: one batch late
            got.append([g.clone() for g in pipe.reduced()])
    while pipe._in_flight:
        got.append([g.clone() for g in pipe.reduced()])
    if rank == 0:
        torch.save(got, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pipelined_exchange_delivers_every_batch_in_order(tmp_path):
    out = str(tmp_path / "pipe.pt")
    port = 29500 + ((os.getpid() + 313) % 500)
    mp.spawn(_pipelined_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    scene, cams = _views_and_scene()
    assert len(got) == 3
    for batch, grads in zip([cams[0:2], cams[2:5], cams[1:4]], got):
        params = _params(scene)
        render = _make_render_fn(params)
        for c in batch:
            render(c).backward()
        for g2, p in zip(grads, params):
            assert (g2 - p.grad).norm().item() <= 1e-6 * max(p.grad.norm().item(), 1e-12)


def test_single_process_passthrough():
    """Without an initialised process group the sharded renderer is the plain loop over all views."""
    from bags_raster.sharding import GradAllReducer, ViewShardedRenderer
    w = torch.ones(3, requires_grad=True)
    r = ViewShardedRenderer([w], lambda v: (w * v).sum())
    res = r.step([1.0, 2.0, 3.0])
    assert res["views"] == [0, 1, 2] and torch.allclose(w.grad, torch.full((3,), 6.0))
    GradAllReducer([w]).all_reduce()      # no-op


def test_flat_bucket_layout_and_absorb():
    """The bucket's layout depends on shapes and world size only; gradients autograd left outside it are absorbed."""
    from bags_raster.sharding import FlatGradBucket
    a, b = torch.zeros(10, 3, requires_grad=True), torch.zeros(7, requires_grad=True)
    bk = FlatGradBucket([a, b], world=8)
    assert bk.offsets == [0, 64] and bk.numel % (64 * 8) == 0 and bk.flat.abs().sum() == 0
    bk.bind()
    (a.sum() * 2 + b.sum() * 3).backward()
    assert bk.bound() and torch.equal(bk.flat[:30], torch.full((30,), 2.0)) and torch.equal(bk.flat[64:71], torch.full((7,), 3.0))
    a.grad = None                                         # a caller's zero_grad(set_to_none=True)
    (a.sum() * 5).backward()                              # autograd now owns a fresh tensor
    assert not bk.bound()
    bk.absorb()
    assert bk.bound() and torch.equal(bk.flat[:30], torch.full((30,), 7.0))
    with pytest.raises(ValueError):
        FlatGradBucket([a, torch.zeros(3, dtype=torch.float64)])


def _stats_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bags_raster.sharding import DensificationSync, shard_views
    pc, views = _stats_case()
    sync = DensificationSync(pc)
    for it in range(2):                                   # two exchange intervals: the increments must not be counted twice
        for v in shard_views(len(views), rank, world):
            _feed(pc, views[v], it)
        sync.sync()
    if rank == 0:
        torch.save({"accum": pc.xyz_gradient_accum, "denom": pc.denom, "radii": pc.max_radii2D}, out)
    dist.destroy_process_group()


def _stats_case():
    from bags_raster.gaussians import GaussianBag
    from bags_raster.synth import synth_scene
    pc = GaussianBag.from_activated(synth_scene(50, 1, 1.0, 0), 0)
    g = torch.Generator().manual_seed(3)
    views = [dict(grad=torch.randn(50, 3, generator=g), seen=torch.rand(50, generator=g) < 0.6,
                  radii=torch.randint(0, 40, (50,), generator=g).float()) for _ in range(5)]
    return pc, views


def _feed(pc, view, it):
    """What train.py does per view: max of the radii of the visible Gaussians, then add_densification_stats."""
    vp = torch.zeros(50, 3, requires_grad=True)
    vp.grad = view["grad"] * (it + 1)
    seen = view["seen"]
    pc.max_radii2D[seen] = torch.max(pc.max_radii2D[seen], view["radii"][seen])
    pc.add_densification_stats(vp, None, seen, abs_grad=False)


@pytest.mark.timeout(120)
def test_densification_stats_match_single_process(tmp_path):
    out = str(tmp_path / "stats.pt")
    port = 29500 + ((os.getpid() + 77) % 500)
    mp.spawn(_stats_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    pc, views = _stats_case()
    for it in range(2):
        for v in views:
            _feed(pc, v, it)
    assert torch.allclose(got["accum"], pc.xyz_gradient_accum, rtol=1e-6, atol=1e-6)
    assert torch.equal(got["denom"], pc.denom) and torch.equal(got["radii"], pc.max_radii2D)



def _sparse_worker(rank, world, port, out):
    """mode="sparse": rows no rank touched stay home.  Rank 0 touches rows 0..39 and 100..119, rank 1 rows 30..59: the union
    (80 of 320 rows) is what crosses the links, the result equals the dense sum bit for bit (same two addends per element);
    with dense_above = 0.2 the same gradients take the dense all-reduce instead."""
    _setup(rank, world, port)
    from bags_raster.sharding import GradAllReducer
    g = torch.Generator().manual_seed(11 + rank)
    P = 320
    shapes = [(P, 3), (P, 3), (P, 4), (P, 1), (P, 4, 3)]
    rows = torch.zeros(P, dtype=torch.bool)
    if rank == 0:
        rows[0:40] = True; rows[100:120] = True
    else:
        rows[30:60] = True
    results = {}
    for tag, dense_above in (("sparse", 0.7), ("fallback", 0.2)):
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        red = GradAllReducer(params, mode="sparse", dense_above=dense_above)
        red.begin()
        local = []
        for p in params:
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(5 + rank + p.dim())) * rows.view(-1, *([1] * (p.dim() - 1)))
            p.grad.add_(gr)
            local.append(gr)
        red.all_reduce()
        results[tag] = dict(grads=[p.grad.clone() for p in params], rows=red.exchange.last_rows_exchanged,
                            collectives=red.exchange.collectives_issued)
        # reference: plain dense sum of both ranks' local gradients
        for p, gr in zip(params, local):
            ref = gr.clone()
            dist.all_reduce(ref)
            assert torch.equal(ref, p.grad), tag
    assert results["sparse"]["rows"] == 80 and results["sparse"]["collectives"] == 2       # mask + compact rows
    assert results["fallback"]["rows"] == P and results["fallback"]["collectives"] == 2     # mask + dense bucket
    if rank == 0:
        torch.save({k: v["rows"] for k, v in results.items()}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_exchange_moves_only_touched_rows(tmp_path):
    out = str(tmp_path / "sparse.pt")
    port = 29500 + ((os.getpid() + 401) % 500)
    mp.spawn(_sparse_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.load(out) == {"sparse": 80, "fallback": 320}


def _single_view_worker(rank, world, port, out):
    """One view per rank per exchange without the bucket: the collective runs on the buffer the op's backward carved its
    gradients from.  Equal to the bucket path bit for bit; a rank whose gradients are separate tensors raises."""
    _setup(rank, world, port)
    from bags_raster.sharding import GradAllReducer
    scene, cams = _views_and_scene()
    res = {}
    for tag in ("bucket", "single"):
        params = _params(scene)
        render = _make_render_fn(params)
        red = GradAllReducer(params)
        if tag == "bucket":
            red.begin()
            render(cams[rank]).backward()
            red.all_reduce()
        else:
            render(cams[rank]).backward()                  # p.grad = the op's carved views, handed over without a copy
            red.all_reduce_single_view()
            assert red.single_view_collectives == 1 and red.exchange.collectives_issued == 0
        res[tag] = [p.grad.clone() for p in params]
    for a, b in zip(res["bucket"], res["single"]):
        assert torch.equal(a, b)
    # several views per rank: the first backward's buffer is adopted as the accumulator (autograd adds the later views into it
    # in place), no bucket zero / first-view add passes -- and still bit for bit what the bucket path gives
    for tag in ("bucket2", "adopt2"):
        params = _params(scene)
        render = _make_render_fn(params)
        red = GradAllReducer(params)
        if tag == "bucket2":
            red.begin()
        for c in (cams[rank], cams[rank + 2]):
            render(c).backward()
        if tag == "bucket2":
            red.all_reduce()
        else:
            red.all_reduce_adopted()
        res[tag] = [p.grad.clone() for p in params]
    for a, b in zip(res["bucket2"], res["adopt2"]):
        assert torch.equal(a, b)
    params = _params(scene)
    for p in params:
        p.grad = torch.zeros_like(p)                       # unrelated tensors: not one buffer
    with pytest.raises(RuntimeError, match="not views of one buffer"):
        GradAllReducer(params).all_reduce_single_view()
    if rank == 0:
        torch.save(res["single"], out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_single_view_exchange_runs_on_the_ops_own_gradient_buffer(tmp_path):
    out = str(tmp_path / "single.pt")
    port = 29500 + ((os.getpid() + 457) % 500)
    mp.spawn(_single_view_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    scene, cams = _views_and_scene()
    params = _params(scene)
    render = _make_render_fn(params)
    for c in cams[:2]:
        render(c).backward()
    for g2, p in zip(got, params):
        assert (g2 - p.grad).norm().item() <= 1e-6 * max(p.grad.norm().item(), 1e-12)
