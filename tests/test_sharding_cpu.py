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


def _views_and_scene():
    from bags_raster.synth import sphere_views, synth_scene
    scene = synth_scene(320, 3, 3.0, 1)                    # 320: every gradient is a multiple of 64 floats (no padding in the flat buffer)
    cams = sphere_views(5, 48, 32, noise=0.05)            # 5 views: uneven split over 2 ranks (3 + 2)
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


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bags_raster.sharding import ViewShardedRenderer, shard_views
    scene, cams = _views_and_scene()
    params = [scene[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    r = ViewShardedRenderer(params, _make_render_fn(params))
    res = r.step(cams)
    assert res["views"] == shard_views(len(cams), rank, world)
    from bags_raster.sharding import coalesce_by_storage
    assert len(coalesce_by_storage([p.grad for p in params])) == 1, "the carved gradients did not reach the reducer as one buffer"
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in params], "loss": res["loss_sum"]}, out)
    # every rank must hold the same summed gradients
    for p in params:
        ref = p.grad.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, p.grad)
    dist.destroy_process_group()


def test_shard_views_round_robin():
    from bags_raster.sharding import shard_views
    assert shard_views(5, 0, 2) == [0, 2, 4] and shard_views(5, 1, 2) == [1, 3]
    assert shard_views(3, 5, 8) == [] and sorted(sum((shard_views(200, r, 8) for r in range(8)), [])) == list(range(200))


@pytest.mark.timeout(300)
def test_two_rank_sum_equals_single_process(tmp_path):
    out = str(tmp_path / "rank0.pt")
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # single process, all 5 views
    scene, cams = _views_and_scene()
    params = [scene[k].clone().requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")]
    render = _make_render_fn(params)
    total = 0.0
    for c in cams:
        loss = render(c)
        loss.backward()
        total += float(loss)
    assert abs(float(got["loss"]) - total) < 1e-5 * max(1.0, abs(total))
    for g2, p in zip(got["grads"], params):
        denom = p.grad.norm().item()
        assert (g2 - p.grad).norm().item() <= 1e-6 * max(denom, 1e-12), "N-rank sum differs from 1-process sum"


def test_single_process_passthrough():
    """Without an initialised process group the sharded renderer is the plain loop over all views."""
    from bags_raster.sharding import GradAllReducer, ViewShardedRenderer
    w = torch.ones(3, requires_grad=True)
    r = ViewShardedRenderer([w], lambda v: (w * v).sum())
    res = r.step([1.0, 2.0, 3.0])
    assert res["views"] == [0, 1, 2] and torch.allclose(w.grad, torch.full((3,), 6.0))
    GradAllReducer([w]).all_reduce()      # no-op


def test_coalesce_by_storage_groups_carved_gradients():
    """Gradients carved out of one buffer are exchanged as one flat tensor; strangers and sparse layouts pass through."""
    from bags_raster.sharding import coalesce_by_storage
    flat = torch.arange(64 * 5, dtype=torch.float32)
    a, b, c = flat[0:30].view(10, 3), flat[64:64 + 40].view(10, 4), flat[128:128 + 192].view(4, 16, 3)
    lone = torch.ones(7)
    out = coalesce_by_storage([a, lone, c, b], max_waste=0.5)
    assert len(out) == 2
    big = max(out, key=lambda t: t.numel())
    assert big.numel() == 128 + 192 and big.data_ptr() == flat.data_ptr()
    big += 1.0                                   # what an in-place all-reduce does
    assert torch.equal(a, (torch.arange(30, dtype=torch.float32) + 1).view(10, 3))
    assert torch.equal(c.reshape(-1), torch.arange(128, 320, dtype=torch.float32) + 1)
    # too much padding between the pieces: left alone
    out = coalesce_by_storage([flat[0:8], flat[300:308]])
    assert len(out) == 2 and all(t.numel() == 8 for t in out)
    # overlapping views of one storage are never merged
    out = coalesce_by_storage([flat[0:16], flat[8:24]])
    assert len(out) == 2


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

