"""The view-sharded exchange (bags_raster/sharding.py) sums ONE persistent flat bucket whose slices are the parameters'
``.grad``: the HIP op's backward (direct call and through render() + GaussianBag's fused activations) must accumulate
into it in place, view after view."""
import pytest
import torch

from scenes import hip_settings, make_case

pytestmark = pytest.mark.gpu


def test_op_gradients_accumulate_in_the_flat_bucket():
    from bags_raster import GaussianRasterizer
    from bags_raster.sharding import GradAllReducer
    dev = torch.device("cuda", 0)
    scene, cam = make_case(3000, 160, 96, 1.5, 3, seed=2)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    P = leaves["means3D"].shape[0]
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, 3, dev))

    def backward_once():
        img = rast(means3D=leaves["means3D"], means2D=m2d, means2D_densify=None, shift_factors=None, shs=leaves["shs"],
                   colors_precomp=None, opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                   cov3D_precomp=None)[0]
        img.sum().backward()
    backward_once()
    ref = [leaves[k].grad.clone() for k in names]             # plain autograd result of one view
    red = GradAllReducer([leaves[k] for k in names])
    red.begin()
    assert red.bucket.bound() and float(red.bucket.flat.abs().sum()) == 0.0
    backward_once()
    backward_once()                                           # V = 2 views behind one exchange
    assert red.bucket.bound(), "autograd replaced a bucket slice instead of accumulating into it"
    for k, r, o in zip(names, ref, red.bucket.offsets):
        assert torch.allclose(leaves[k].grad, 2.0 * r, rtol=1e-5, atol=1e-6), k
        assert leaves[k].grad.data_ptr() == red.bucket.flat.data_ptr() + 4 * o
    red.all_reduce()                                          # no process group: a no-op, gradients untouched
    assert torch.allclose(leaves["means3D"].grad, 2.0 * ref[0], rtol=1e-5, atol=1e-6)
    red.begin()                                               # next iteration starts from zeros
    assert float(red.bucket.flat.abs().sum()) == 0.0


def test_render_path_raw_leaves_accumulate_in_the_flat_bucket():
    """render() + GaussianBag: the raw leaves' gradients come out of the fused-activation backward as tensors of their
    own; with the bucket bound they are added into its slices (ONE collective for the exchange)."""
    from bags_raster.gaussians import GaussianBag
    from bags_raster.render import PipelineParams, render
    from bags_raster.sharding import GradAllReducer
    from bags_raster.synth import sphere_views
    dev = "cuda"
    scene, _ = make_case(1500, 160, 128, 1.5, 3, seed=17)
    cams = sphere_views(3, 160, 128, noise=0.05, device=dev)
    bg = torch.zeros(3, device=dev)
    g = torch.randn(3, 128, 160, generator=torch.Generator().manual_seed(3)).to(dev)
    pc = GaussianBag.from_activated(scene, 3, device=dev)
    for c in cams:
        render(c, pc, PipelineParams(), bg, 0.0, None, hybrid=False)["render"].backward(g)
    ref = [p.grad.clone() for p in pc.leaves()]
    pc2 = GaussianBag.from_activated(scene, 3, device=dev)
    red = GradAllReducer(pc2.leaves())
    red.begin()
    for c in cams:
        render(c, pc2, PipelineParams(), bg, 0.0, None, hybrid=False)["render"].backward(g)
    assert red.bucket.bound()
    for p, r in zip(pc2.leaves(), ref):
        assert torch.allclose(p.grad, r, rtol=1e-5, atol=1e-6)
    assert red.exchange.collectives_issued == 0               # single process: nothing to exchange


def test_five_views_with_distance_depth_key_accumulate_to_the_oracle_sum():
    """The cubemap step of the reference issues five op calls per iteration with the Euclidean-distance sort key
    (utils/cubemap_utils.py:229,263-265, README.md:126) and backpropagates their summed loss.  Five views through
    ViewShardedRenderer.step (one process: V = 5 views behind one exchange point), depth_key="distance": the gradients
    accumulated in the flat bucket must equal the sum of the oracle's five per-view gradients."""
    from bags_raster import GaussianRasterizer
    from bags_raster.sharding import ViewShardedRenderer
    from bags_raster.synth import sphere_views
    from parity import run_oracle
    from scenes import rel_err
    dev = torch.device("cuda", 0)
    W, H, deg = 128, 96, 2
    scene, _ = make_case(2500, W, H, 1.5, deg, seed=61)
    cams = sphere_views(5, W, H, noise=0.1, seed=7)
    cots = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(70 + k)) for k in range(5)]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    leaves = {k: scene[k].to(dev).clone().requires_grad_(True) for k in names}
    P = scene["means3D"].shape[0]

    def render_fn(view):
        k, cam = view
        st = hip_settings(cam, deg, dev, depth_key="distance")
        img = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=torch.zeros(P, 3, device=dev), means2D_densify=None,
                                     shift_factors=None, shs=leaves["shs"], colors_precomp=None, opacities=leaves["opacities"],
                                     scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)[0]
        return (img * cots[k].to(dev)).sum()
    r = ViewShardedRenderer([leaves[k] for k in names], render_fn)
    res = r.step(list(enumerate(cams)))
    assert res["views"] == [0, 1, 2, 3, 4] and r.reducer.bucket.bound()
    want = {k: torch.zeros_like(scene[k]) for k in names}
    for k, cam in enumerate(cams):
        _, gr = run_oracle(scene, cam, deg, cots[k], torch.float32, depth_key="distance")
        for n in names:
            want[n] += gr[n]
    for n in names:
        assert rel_err(leaves[n].grad.cpu(), want[n]) < 1e-4, (n, rel_err(leaves[n].grad.cpu(), want[n]))


@pytest.mark.gpu
def test_views_accumulate_in_place_like_autograd():
    """rasterizer.ACCUMULATE_IN_PLACE (opt-in): with a gradient already in every Gaussian parameter, the backward of the next view
    adds into those tensors inside the kernel (BagsBackwardArgs.accumulate) and returns None to autograd.  old + new is the same
    fp32 addition autograd's accumulation performs: bit-identical sums; per-view outputs (pose tensors, means2D) are untouched by
    the switch; the first view (no gradient yet) takes the ordinary path."""
    import math
    from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, rasterizer as R
    from bags_raster.synth import sphere_views, synth_scene
    from scenes import camera_tensors
    dev = torch.device("cuda")
    P, W, H, deg = 3000, 160, 112, 2
    scene = synth_scene(P, 7, 1.5, deg)
    cams = sphere_views(3, W, H, noise=0.05)
    cots = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(10 + v)).to(dev) for v in range(3)]

    def run(in_place):
        R.ACCUMULATE_IN_PLACE = in_place
        try:
            leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
            per_view = []
            for cam, cot in zip(cams, cots):
                ct = {k: v.clone().requires_grad_(True) for k, v in camera_tensors(cam, dev).items()}
                m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
                st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                                   tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                                   viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], intrinsic=ct["intrinsic"],
                                                   sh_degree=deg, campos=ct["campos"])
                img = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
                                             scales=leaves["scales"], rotations=leaves["rotations"])[0]
                img.backward(cot)
                per_view.append({**{k: v.grad.clone() for k, v in ct.items()}, "means2D": m2.grad.clone()})
            return {k: v.grad.clone() for k, v in leaves.items()}, per_view
        finally:
            R.ACCUMULATE_IN_PLACE = False
    g_ref, pv_ref = run(False)
    g_acc, pv_acc = run(True)
    for k in g_ref:
        assert torch.equal(g_ref[k], g_acc[k]), k
    for a, b in zip(pv_ref, pv_acc):
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams,factored", [(1, False), (2, False), (3, False), (2, True)])
def test_views_on_several_streams_accumulate_in_view_order(n_streams, factored):
    """ABI 9: bags_backward in two halves (BagsBackwardArgs.phase) + rasterizer.AccumulationGate.  The views of one step run on
    n_streams streams with their gradients accumulating in place; the per-Gaussian half of every backward waits for the one before
    it, so the sums are formed in the order the backwards were CALLED in -- bit for bit what the same calls give on ONE stream without a
    gate, and what autograd's own accumulation gives (test_views_accumulate_in_place_like_autograd).  n_streams = 1: the two-halves
    path alone (BAGS_BWD_BLEND, then BAGS_BWD_PREPROCESS) against the one-call backward."""
    import math
    from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, rasterizer as R
    from bags_raster.synth import sphere_views, synth_scene
    from scenes import camera_tensors
    dev = torch.device("cuda")
    P, W, H, deg, V = 5000, 256, 176, 3, 4
    scene = synth_scene(P, 9, 1.5, deg)
    cams = sphere_views(V, W, H, noise=0.05)
    cots = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(20 + v)).to(dev) for v in range(V)]
    base = {k: v.to(dev) for k, v in scene.items()}

    def run(streams, gate, factored_sh=False):
        saved = (R.ACCUMULATE_IN_PLACE, R.ACCUMULATION_GATE, R.HOST_WAIT)
        R.ACCUMULATE_IN_PLACE, R.ACCUMULATION_GATE = True, gate
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            cur = torch.cuda.current_stream()
            per_view = []
            for rep in range(3):                              # (the second call of a shape on takes the speculative forward)
                if gate is not None:
                    gate.reset()
                for p in leaves.values():
                    p.grad = None
                per_view = []
                fs = R.FactoredSH() if factored_sh else None         # (with the gate: finish() waits for the views' streams itself)
                R.FACTORED_SH = fs
                for s in streams:
                    s.wait_stream(cur)
                for v, (cam, cot) in enumerate(zip(cams, cots)):
                    with torch.cuda.stream(streams[v % len(streams)]):
                        ct = {k: t.clone().requires_grad_(True) for k, t in camera_tensors(cam, dev).items()}
                        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
                        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                                           tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev),
                                                           scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"],
                                                           intrinsic=ct["intrinsic"], sh_degree=deg, campos=ct["campos"])
                        img = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
                                                     scales=leaves["scales"], rotations=leaves["rotations"])[0]
                        img.backward(cot)
                        per_view.append((ct, m2))
                R.FACTORED_SH = None
                if fs is not None:
                    fs.finish(leaves["means3D"], leaves["shs"])
                for s in streams:
                    cur.wait_stream(s)
                torch.cuda.synchronize()
            return ({k: p.grad.clone() for k, p in leaves.items()},
                    [{**{k: t.grad.clone() for k, t in ct.items()}, "means2D": m2.grad.clone()} for ct, m2 in per_view])
        finally:
            R.ACCUMULATE_IN_PLACE, R.ACCUMULATION_GATE, R.HOST_WAIT = saved
            R.FACTORED_SH = None
    g_ref, pv_ref = run([torch.cuda.current_stream()], None)
    streams = [torch.cuda.current_stream()] if n_streams == 1 else [torch.cuda.Stream() for _ in range(n_streams)]
    g_new, pv_new = run(streams, R.AccumulationGate(), factored_sh=factored)
    for k in g_ref:
        assert torch.equal(g_ref[k], g_new[k]), k
    for a, b in zip(pv_ref, pv_new):
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_backward_twice_on_one_forward_and_forward_without_backward():
    """rasterizer.PREALLOCATE_BACKWARD (round 6): a waiting forward allocates its backward's buffers while it waits for the instance
    count.  They belong to the FIRST backward of that forward: a second backward on the same graph (retain_graph) must get buffers of
    its own (the first call's tensors are the caller's gradients by then), with the same values; a forward that is never
    differentiated, and one under no_grad, simply give the buffers back; the switch itself does not change a bit."""
    import math
    from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, rasterizer as R
    from bags_raster.synth import look_at_origin_camera, synth_scene
    from scenes import camera_tensors
    dev = torch.device("cuda")
    P, W, H, deg = 4000, 200, 136, 2
    scene = synth_scene(P, 3, 1.5, deg)
    cam = look_at_origin_camera(W, H)
    cot = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)

    def forward(leaves, ct):
        st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                           bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                           projmatrix=ct["projmatrix"], intrinsic=ct["intrinsic"], sh_degree=deg, campos=ct["campos"])
        return GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=torch.zeros(P, 3, device=dev, requires_grad=True), shs=leaves["shs"],
                                      opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"])[0]

    def grads(prealloc, twice):
        saved = R.PREALLOCATE_BACKWARD
        R.PREALLOCATE_BACKWARD = prealloc
        try:
            out = None
            for _ in range(2):                                   # (the second call of a shape takes the speculative, waiting forward)
                leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
                ct = {k: v.clone().requires_grad_(True) for k, v in camera_tensors(cam, dev).items()}
                img = forward(leaves, ct)
                ins = list(leaves.values()) + list(ct.values())
                g1 = torch.autograd.grad(img, ins, cot, retain_graph=twice)
                g2 = torch.autograd.grad(img, ins, cot) if twice else g1
                out = (g1, g2)
            return out
        finally:
            R.PREALLOCATE_BACKWARD = saved
    (a1, a2), (b1, _), (c1, _) = grads(True, True), grads(True, False), grads(False, False)
    for x, y, z, w in zip(a1, a2, b1, c1):
        assert torch.equal(x, y) and torch.equal(x, z) and torch.equal(x, w)
        assert x.data_ptr() != y.data_ptr()                      # the second backward did not write into the first one's result
    # never differentiated / no_grad: nothing to assert but that they run and leave the next differentiated call intact
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
    ct = {k: v.clone().requires_grad_(True) for k, v in camera_tensors(cam, dev).items()}
    img = forward(leaves, ct)
    with torch.no_grad():
        img2 = forward(leaves, ct)
    assert torch.equal(img.detach(), img2)
    del img, img2
    (d1, _) = grads(True, False)
    for x, y in zip(a1, d1):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("M,deg,split,P", [(16, 3, False, 3001), (16, 3, True, 3001), (16, 1, True, 700), (9, 2, False, 515), (4, 1, True, 130)])
def test_factored_sh_gradient_equals_the_accumulated_one(M, deg, split, P):
    """ABI 10: rasterizer.FactoredSH.  Every view's backward writes only dL/dcolour (P,3); finish() forms the SH-gradient rows of the
    whole step in one kernel.  Same products (basis at the view's direction x dL/dcolour), added in view order: the result must be
    bit for bit what the same backwards leave with ACCUMULATE_IN_PLACE (and with autograd's own accumulation) -- for the concatenated
    (P,M,3) tensor and for the features_dc / features_rest pair, M = 16 (whole-line kernel) and smaller M (row kernel), an active
    degree below the stored one, P not a multiple of the workgroup's 128 Gaussians, a second step accumulating on top of the first;
    every other gradient untouched by the switch."""
    import math
    from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, rasterizer as R
    from bags_raster.synth import sphere_views, synth_scene
    from scenes import camera_tensors
    dev = torch.device("cuda")
    W, H, V = 176, 128, 3
    g = torch.Generator().manual_seed(M * 100 + deg)
    scene = synth_scene(P, 11, 1.5, 3)
    scene["shs"] = (torch.randn(P, M, 3, generator=g) * 0.3)
    cams = sphere_views(V, W, H, noise=0.05)
    cots = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(30 + v)).to(dev) for v in range(V)]

    def run(factored, steps=2):
        saved = (R.ACCUMULATE_IN_PLACE, R.FACTORED_SH)
        R.ACCUMULATE_IN_PLACE = True
        try:
            leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items() if k != "shs"}
            if split:
                dc = scene["shs"][:, :1].contiguous().to(dev).requires_grad_(True)
                rest = scene["shs"][:, 1:].contiguous().to(dev).requires_grad_(True)
            else:
                dc, rest = scene["shs"].to(dev).clone().requires_grad_(True), None
            per_view = []
            for step in range(steps):                           # the second step accumulates on top of the first one's gradients
                fs = R.FactoredSH() if factored else None
                R.FACTORED_SH = fs
                for cam, cot in zip(cams, cots):
                    ct = {k: t.clone().requires_grad_(True) for k, t in camera_tensors(cam, dev).items()}
                    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                                       projmatrix=ct["projmatrix"], intrinsic=ct["intrinsic"], sh_degree=deg, campos=ct["campos"])
                    img = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=torch.zeros(P, 3, device=dev, requires_grad=True), shs=dc, shs_rest=rest,
                                                 opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"])[0]
                    img.backward(cot)
                    per_view.append({k: t.grad.clone() for k, t in ct.items()})
                R.FACTORED_SH = None
                if fs is not None:
                    assert dc.grad is None or step > 0                       # nothing reaches the SH parameters before finish()
                    fs.finish(leaves["means3D"], dc, rest)
            out = {k: v.grad.clone() for k, v in leaves.items()}
            out["dc"] = dc.grad.clone()
            if rest is not None:
                out["rest"] = rest.grad.clone()
            return out, per_view
        finally:
            R.ACCUMULATE_IN_PLACE, R.FACTORED_SH = saved
    g_ref, pv_ref = run(False)
    g_fac, pv_fac = run(True)
    for k in g_ref:
        assert torch.equal(g_ref[k], g_fac[k]), (k, float((g_ref[k] - g_fac[k]).abs().max()))
    for a, b in zip(pv_ref, pv_fac):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert float(g_ref["dc"].abs().max()) > 0


@pytest.mark.gpu
def test_view_sharded_step_with_factored_sh_equals_the_plain_step():
    """ViewShardedRenderer(sh_params=...): the rank's views keep their SH gradients factored and one pass forms the step's rows in the
    bucket's slices before the exchange.  Single process (the exchange is a no-op): every gradient in the bucket bit-identical to the
    same step without sh_params, the bucket still bound (ONE collective would follow)."""
    from bags_raster import GaussianRasterizer
    from bags_raster.sharding import ViewShardedRenderer
    from bags_raster.synth import sphere_views
    dev = torch.device("cuda", 0)
    scene, _ = make_case(4000, 192, 128, 1.5, 3, seed=23)
    cams = sphere_views(3, 192, 128, noise=0.05)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    cot = torch.randn(3, 128, 192, generator=torch.Generator().manual_seed(4)).to(dev)

    def run(factored):
        leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in scene.items()}
        P = leaves["means3D"].shape[0]

        def render_fn(cam):
            img = GaussianRasterizer(hip_settings(cam, 3, dev))(
                means3D=leaves["means3D"], means2D=torch.zeros(P, 3, device=dev, requires_grad=True), means2D_densify=None, shift_factors=None,
                shs=leaves["shs"], colors_precomp=None, opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                cov3D_precomp=None)[0]
            return (img * cot).sum()
        r = ViewShardedRenderer([leaves[k] for k in names], render_fn,
                                sh_params=(leaves["means3D"], leaves["shs"]) if factored else None)
        for _ in range(2):                                          # second step: the bucket is zeroed and bound again
            out = r.step(cams)
        assert r.reducer.bucket.bound() and out["views"] == [0, 1, 2]
        return {k: leaves[k].grad.clone() for k in names}
    a, b = run(False), run(True)
    for k in names:
        assert torch.equal(a[k], b[k]), k
