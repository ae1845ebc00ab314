"""Host-side mirror of the reference's rasterizer operator (the Python half of ``diff_gaussian_rasterization``).

Reference interface reproduced here (the fork's own source is absent, so the contract is its call site):
  * ``GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
    projmatrix, intrinsic, sh_degree, campos, prefiltered, debug, debug_iter)``   gaussian_renderer/__init__.py:50-65
  * ``GaussianRasterizer(raster_settings)(means3D=, means2D=, means2D_densify=, shift_factors=, shs=,
    colors_precomp=, opacities=, scales=, rotations=, cov3D_precomp=)`` -> ``(rendered_image, radii, depth, weights,
    mean2D)``                                                                      gaussian_renderer/__init__.py:110-121
  * gradients flow to the Gaussian parameters AND to the tensors inside the settings (viewmatrix, projmatrix,
    intrinsic, campos) plus shift_factors: that is how train.py:472-485 optimises the camera leaves.

PyTorch is plumbing only: it owns device memory and the stream.  All compute is in libbags_raster.so through the
C ABI (``_lib``); tensors must live on the AMD GPU ("cuda" device under ROCm).  There is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import collections
import math
import threading
import time
import weakref

import torch

from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    intrinsic: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False
    debug_iter: Optional[int] = None
    depth_key: str = "z"            # "distance" replaces the README.md:126 hand-edit + recompile for cubemaps
    # "opacity": bin a Gaussian only into tiles its alpha >= 1/255 ellipse can reach (same image / gradients, fewer
    # sorted instances); "aabb": the stock 3-sigma square of upstream 3DGS (identical instance lists to the CUDA code)
    tile_bounds: str = "opacity"
    # "auto": per-tile lists from the (block, tile) count matrix + per-tile LDS sort (csrc/binning.hip) when the image has
    # <= 32768 tiles; "radix": depth sort of the Gaussians + stable radix sort of the instances (csrc/sort.hip).  Same lists.
    binning: str = "auto"
    # gradient of the EWA Jacobian at the 1.3 x FoV clamp: "stock" = upstream diff-gaussian-rasterization's rule (clamped t.x
    # held constant inside dL/dt.z), which the reference's fork inherits; "exact" differentiates the clamped expression itself
    clamp_grad: str = "stock"
    # backward of conic = cov2D^-1: "stock" = upstream's computeCov2DCUDA, which divides by det^2 + 1e-7 (the fork inherits it;
    # default since round 5); "exact" = det^2.  At most 1.2e-5 relative on one Gaussian's dL/dcov2D (det >= 0.09)
    conic_grad: str = "stock"


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: Optional[torch.Tensor], name: str, device: torch.device, shape=None, empty_ok: bool = False) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a tensor")
    if t.numel() == 0 and shape is None and not (empty_ok and t.dim() > 1):
        return None                       # the fork's wrapper passes torch.Tensor([]) for "absent"
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


class _Packed:
    """C structs + the tensors that keep their pointers alive."""

    def __init__(self, settings: GaussianRasterizationSettings, means3D, means2D, shift_factors, sh, colors_precomp,
                 opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, intrinsic, campos, sh_rest=None):
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("bags_raster runs only on an AMD GPU: tensors must be on a 'cuda' (ROCm) device; "
                               "there is no CPU path")
        P = means3D.shape[0]
        self.device, self.P = dev, P
        k = {}
        k["means3D"] = _f32c(means3D, "means3D", dev, (P, 3))
        k["means2D"] = _f32c(means2D, "means2D", dev, (P, 3)) if means2D is not None else None
        k["shift_factors"] = _f32c(shift_factors, "shift_factors", dev, (3,)) if shift_factors is not None else None
        e0 = P == 0                          # an empty scene: (0, n) tensors are real arguments, not the "absent" marker
        k["shs"] = _f32c(sh, "shs", dev, empty_ok=e0)
        k["shs_rest"] = _f32c(sh_rest, "shs_rest", dev, empty_ok=e0)
        k["colors_precomp"] = _f32c(colors_precomp, "colors_precomp", dev, empty_ok=e0)
        k["opacities"] = _f32c(opacities, "opacities", dev, empty_ok=e0)
        k["scales"] = _f32c(scales, "scales", dev, empty_ok=e0)
        k["rotations"] = _f32c(rotations, "rotations", dev, empty_ok=e0)
        k["cov3D_precomp"] = _f32c(cov3D_precomp, "cov3D_precomp", dev, empty_ok=e0)
        if (k["shs"] is None) == (k["colors_precomp"] is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((k["scales"] is None or k["rotations"] is None) and k["cov3D_precomp"] is None) or \
                ((k["scales"] is not None or k["rotations"] is not None) and k["cov3D_precomp"] is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        M = 0
        if k["shs"] is not None:
            if k["shs"].dim() != 3 or k["shs"].shape[0] != P or k["shs"].shape[2] != 3:
                raise ValueError(f"shs must be (P,M,3), got {tuple(k['shs'].shape)}")
            M = k["shs"].shape[1]
        if k["shs_rest"] is not None:            # the reference's two feature parameters as they are stored: features_dc + features_rest
            if k["shs"] is None or M != 1:
                raise ValueError("shs_rest (features_rest, (P,M-1,3)) goes with shs = features_dc of shape (P,1,3)")
            if k["shs_rest"].dim() != 3 or k["shs_rest"].shape[0] != P or k["shs_rest"].shape[2] != 3 or k["shs_rest"].shape[1] < 1:
                raise ValueError(f"shs_rest must be (P,M-1,3) with M >= 2, got {tuple(k['shs_rest'].shape)}")
            M = 1 + k["shs_rest"].shape[1]
        if k["colors_precomp"] is not None and tuple(k["colors_precomp"].shape) != (P, 3):
            raise ValueError("colors_precomp must be (P,3)")
        if k["opacities"] is None or k["opacities"].numel() != P:
            raise ValueError("opacities must be (P,1)")
        for nm, n in (("scales", 3), ("rotations", 4), ("cov3D_precomp", 6)):
            if k[nm] is not None and tuple(k[nm].shape) != (P, n):
                raise ValueError(f"{nm} must be (P,{n})")
        k["bg"] = _f32c(settings.bg, "bg", dev, (3,))
        k["viewmatrix"] = _f32c(viewmatrix, "viewmatrix", dev, (4, 4))
        k["projmatrix"] = _f32c(projmatrix, "projmatrix", dev, (4, 4))
        k["intrinsic"] = _f32c(intrinsic, "intrinsic", dev, (4, 4))
        k["campos"] = _f32c(campos.reshape(3), "campos", dev, (3,))
        self.keep = k
        self.M = M
        if settings.depth_key not in ("z", "distance"):
            raise ValueError("depth_key must be 'z' or 'distance'")
        if settings.tile_bounds not in ("opacity", "aabb"):
            raise ValueError("tile_bounds must be 'opacity' or 'aabb'")
        if settings.binning not in ("auto", "radix"):
            raise ValueError("binning must be 'auto' or 'radix'")
        if settings.clamp_grad not in ("stock", "exact"):
            raise ValueError("clamp_grad must be 'stock' or 'exact'")
        if settings.conic_grad not in ("stock", "exact"):
            raise ValueError("conic_grad must be 'stock' or 'exact'")
        self.settings = L.BagsSettings(
            int(settings.image_height), int(settings.image_width), float(settings.tanfovx), float(settings.tanfovy),
            float(settings.scale_modifier), int(settings.sh_degree), int(M),
            L.DEPTH_DISTANCE if settings.depth_key == "distance" else L.DEPTH_Z, int(bool(settings.debug)),
            int(settings.debug_iter) if settings.debug_iter is not None else -1,
            L.TILES_OPACITY if settings.tile_bounds == "opacity" else L.TILES_AABB,
            L.BINNING_RADIX if settings.binning == "radix" else L.BINNING_AUTO,
            L.CLAMP_GRAD_EXACT if settings.clamp_grad == "exact" else L.CLAMP_GRAD_STOCK,
            L.CONIC_GRAD_EXACT if settings.conic_grad == "exact" else L.CONIC_GRAD_STOCK,
            _ptr(k["bg"]), _ptr(k["viewmatrix"]), _ptr(k["projmatrix"]), _ptr(k["intrinsic"]), _ptr(k["campos"]))
        self.inputs = L.BagsInputs(P, _ptr(k["means3D"]), _ptr(k["means2D"]), _ptr(k["shift_factors"]), _ptr(k["shs"]),
                                   _ptr(k["colors_precomp"]), _ptr(k["opacities"]), _ptr(k["scales"]),
                                   _ptr(k["rotations"]), _ptr(k["cov3D_precomp"]), _ptr(k["shs_rest"]))


def _require_gpu(t: torch.Tensor) -> None:
    if not torch.is_tensor(t) or t.device.type != "cuda":
        raise RuntimeError("bags_raster runs only on an AMD GPU: tensors must be on a 'cuda' (ROCm) device; "
                           "there is no CPU path")


def _bytes(n: int, device) -> torch.Tensor:
    return torch.empty(int(n), dtype=torch.uint8, device=device)


LAST_NUM_RENDERED = 0      # instance count of the most recent forward whose count has been read (bench/diagnostics)


class _Forwarded:
    """Everything backward (and the parity tests) need from one forward call."""
    __slots__ = ("packed", "geom", "binning", "image", "num_rendered", "capacity", "H", "W", "pending", "outs", "stream",
                 "key", "pre", "waiting", "__weakref__")


# Speculative forward (bags_forward_prepare_async + bags_forward_finish_speculative): the instance counts of earlier calls
# with the same problem shape give the capacity guess for the next one, so both phases of the forward are enqueued back to
# back and the device never waits for the host in the middle of a forward.
SPECULATE = True
# When the host reads the asynchronous instance count of a speculative forward:
#   "forward"  (default) every forward reads its count before it returns (the device is meanwhile busy with the second
#              phase) and redoes an overflowing second phase at once on an exact buffer: the tensors a forward returns are
#              ALWAYS the true render, as the reference's are (train.py:250-331: render -> loss -> backward).
#   "lazy"     opt-in.  A forward that will be differentiated returns WITHOUT waiting; the count is read at the entry of
#              its backward.  Nothing in the forward blocks the host, so the views of a batch can be enqueued back to back,
#              also on several streams (utils/cubemap_utils.py:229,263-265 renders five per iteration).  The buffer is sized
#              CAPACITY_HEADROOM x the largest count seen for the shape.  Should a count still exceed it, the image THAT
#              forward returned had every tile rendered empty and whatever was computed from it (the loss, hence the
#              cotangent) is wrong: the backward RAISES ``SpeculationOverflow`` so that the caller can redo the step
#              (the hint has been raised by then, the retry fits).  With LAZY_RECOVER = True it instead redoes the second
#              phase exactly, writes the true image into the tensors the forward returned, computes the gradients of the
#              true render for the cotangent that was passed in, and issues a RuntimeWarning.  A lazy forward that is
#              never differentiated reports an overflow as a RuntimeWarning when its count is collected.
HOST_WAIT = "forward"
LAZY_RECOVER = False
# Opt-in for loops that differentiate several views of one step into the same parameters (view sharding, the reference's cubemap
# step renders five per iteration, utils/cubemap_utils.py:229,263-265): when every Gaussian-parameter input of the op is a leaf
# that already HOLDS a gradient (contiguous fp32 of its own shape), the backward adds into those tensors itself
# (BagsBackwardArgs.accumulate) and hands autograd None for them -- no flat buffer, no add pass per tensor and view.  The sums are
# what autograd's own accumulation gives; tensor hooks on those parameters do not see the per-view gradients.
ACCUMULATE_IN_PLACE = False


# A forward that waits for its instance count (HOST_WAIT = "forward") has nothing to do while it waits: K1, the prefix and the emission
# have to run before the count exists (~85 us at 500 k Gaussians).  With this switch it allocates, in that wait, what its BACKWARD will
# need (the gradient buffers and the record workspace, sized for the capacity the forward was enqueued with) and keeps them in the
# context: the host's path from "count has arrived" to "blend_bwd launched" -- which the device covers with blend_fwd alone -- loses
# its ~15 allocator calls (~60 us of host time; on the pool's hosts that path was already shorter than blend_fwd, so the step time did not
# move: profiles/r06/ab_round6.txt 7 -- it is insurance for slower hosts).  Nothing else changes; a forward that is never differentiated
# hands the buffers back when its context dies.
PREALLOCATE_BACKWARD = True


class AccumulationGate:
    """Orders the per-Gaussian halves of the backwards of ONE optimisation step whose views run on SEVERAL streams.

    With ACCUMULATE_IN_PLACE the backward of view k reads, adds to and writes the gradient buffers view k-1 wrote: a read-modify-write
    that two streams must not interleave, and whose ORDER decides the rounding of the sums.  While a gate is installed
    (``rasterizer.ACCUMULATION_GATE = AccumulationGate()``; the caller that spreads the views over streams installs it --
    tools/bench_cu_partition.py, tests/test_flat_grads_gpu.py; ViewShardedRenderer keeps its views on one stream, where no gate is
    needed: DESIGN.md section 6 has the measurements that decided that) every backward runs as two calls of the library -- BAGS_BWD_BLEND (the per-tile half, ~85 % of the time, no shared state), then
    BAGS_BWD_PREPROCESS behind the event the previous backward recorded after ITS second half -- so the sums are formed in the order
    the backwards were CALLED in, i.e. bit for bit what the same calls give on one stream, while everything else of the views
    overlaps freely.  ``reset()`` at the start of a step (nothing to wait for)."""

    def __init__(self):
        self.event = None

    def reset(self) -> None:
        self.event = None


ACCUMULATION_GATE: Optional[AccumulationGate] = None


class FactoredSH:
    """The SH gradients of the views of ONE optimisation step, kept factored until the step's last backward has run (ABI 10).

    dL/dshs of one view is, per Gaussian, the outer product of the SH basis at the direction from that view's camera and the
    three floats dL/dcolour -- and writing that 192-byte row (and, when the views accumulate, reading it first) is two thirds of the
    traffic of a backward's per-Gaussian half.  While an instance is installed (``rasterizer.FACTORED_SH = FactoredSH()``) every
    backward of the SH colour path writes only its (P,3) dL/dcolour and hands autograd None for ``shs`` / ``shs_rest``;
    ``finish(means3D, shs[, shs_rest])`` then forms the rows of all the step's views in one kernel -- 12 bytes read per Gaussian
    and view, every row written once -- into ``shs.grad`` (added, in view order, to a gradient that is already there: exactly the
    sums, bit for bit, that the same backwards leave with ACCUMULATE_IN_PLACE).  Tensor hooks on the SH parameters see nothing
    before ``finish``.  One instance per step; ``finish`` clears it for the next."""

    def __init__(self):
        self.views = []            # (campos (3,), dL/dcolour (P,3), active sh degree, stream) per backward, in call order
        self.target = None         # (shs slice, shs_rest slice) of the first backward's gradient buffer, lent for the finished rows

    def finish(self, means3D: torch.Tensor, shs: torch.Tensor, shs_rest: Optional[torch.Tensor] = None) -> None:
        views, self.views = self.views, []
        lent, self.target = self.target, None
        if not views:
            return
        lib = L.load()
        dev, P = means3D.device, means3D.shape[0]
        deg = views[0][2]
        if any(v[2] != deg for v in views):
            raise RuntimeError("FactoredSH.finish: the views of one step were rendered with different sh_degree")
        M = shs.shape[1] + (shs_rest.shape[1] if shs_rest is not None else 0)
        targets = []
        for j, p_ in enumerate((shs, shs_rest)):
            if p_ is None:
                targets.append(None)
                continue
            fresh = p_.grad is None
            if fresh:
                slot = lent[j] if lent is not None else None
                p_.grad = slot if (slot is not None and tuple(slot.shape) == tuple(p_.shape)) else torch.empty_like(p_, memory_format=torch.contiguous_format)
            if p_.grad.dtype != torch.float32 or not p_.grad.is_contiguous() or p_.grad.device != dev:
                raise RuntimeError("FactoredSH.finish: the SH parameters' gradients must be contiguous fp32 tensors on the op's device")
            targets.append((p_.grad, fresh))
        if shs_rest is not None and targets[0][1] != targets[1][1]:
            raise RuntimeError("FactoredSH.finish: shs and shs_rest must both hold a gradient, or neither")
        accumulate = 0 if targets[0][1] else 1
        m3 = means3D.detach()
        if m3.dtype != torch.float32 or not m3.is_contiguous():
            m3 = m3.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            for v in views:                                   # the views' backwards may have run on other streams
                if v[3] != cur:
                    cur.wait_stream(v[3])
            for b in range(0, len(views), L.MAX_SH_VIEWS):
                part = views[b:b + L.MAX_SH_VIEWS]
                sv = L.BagsShViews()
                sv.n_views = len(part)
                for j, v in enumerate(part):
                    sv.campos[j], sv.dldc[j] = v[0].data_ptr(), v[1].data_ptr()
                L.check(lib.bags_sh_gradient_from_views(P, M, deg, m3.data_ptr(), C.byref(sv), targets[0][0].data_ptr(),
                                                        None if targets[1] is None else targets[1][0].data_ptr(), accumulate, cur.cuda_stream),
                        "bags_sh_gradient_from_views")
                accumulate = 1
                for v in part:                                # the factored tensors are read on `cur`: tell their allocator streams
                    if v[3] != cur:
                        v[1].record_stream(cur)


FACTORED_SH: Optional[FactoredSH] = None
# BagsBackwardArgs.dense_per_tile: 0 = the library's threshold (instances per tile, scene average) for the backward's dense-scene
# mode (a byte per gradient record instead of zero records), < 0 never, > 0 that threshold.  Results do not depend on it;
# tools/fuzz_paths.py forces both paths with it.
DENSE_PER_TILE = 0
CAPACITY_HEADROOM = 4.0      # lazy forwards only; a waiting forward sizes for 1.2 x the largest count seen and redoes on overflow
_HINT_KEYS_MAX = 64
_capacity_hint = {}          # (device index, P, W, H) -> largest instance count seen for the shape (a hint only); insertion order = LRU
_below_half = {}             # ... -> consecutive calls whose count stayed below half of it
_hint_lock = threading.RLock()


class SpeculationOverflow(RuntimeError):
    """A lazy forward (HOST_WAIT = 'lazy') held more instances than its speculative buffer: the image it returned was empty."""


def _note_count(key, n: int) -> None:
    """The hint is the MAXIMUM seen (the views of a scene differ by tens of per cent); it only comes down after 256 calls in a
    row below half of it.  It must not drift from call to call: the capacity is a buffer size, and sizes that never repeat
    defeat torch's caching allocator (measured: 13-71 hipMalloc per 40 steps with a 2 % decay per call, none without).
    Shapes that have not been rendered for a while are evicted (densification changes P every few hundred iterations)."""
    n = int(n)
    with _hint_lock:
        old = _capacity_hint.pop(key, 0)
        if n >= old:
            new, _below_half[key] = n, 0
        elif 2 * n < old:
            new = old
            _below_half[key] = _below_half.get(key, 0) + 1
            if _below_half[key] >= 256:
                new, _below_half[key] = 2 * n, 0
        else:
            new, _below_half[key] = old, 0
        _capacity_hint[key] = new                              # re-inserted last: most recently used
        while len(_capacity_hint) > _HINT_KEYS_MAX:
            oldest = next(iter(_capacity_hint))
            del _capacity_hint[oldest]
            _below_half.pop(oldest, None)


def _capacity_for(hint: int, headroom: float) -> int:
    """headroom x hint, rounded up to 1/8-octave steps so that the buffer sizes of a run repeat."""
    want = int(hint * headroom) + 8192
    step = 1 << max(10, want.bit_length() - 4)
    return (want + step - 1) // step * step


class _PinnedSlots:
    """Pinned int32 words that receive the asynchronous instance count.  One word per call in flight: two forwards on
    different streams (or threads) of one device must not share the word their counts are copied into.  A slot goes back
    to the pool once its call has read it."""

    def __init__(self):
        self._free, self._lock = [], threading.Lock()

    def take(self) -> torch.Tensor:
        with self._lock:
            if self._free:
                return self._free.pop()
        return torch.zeros(1, dtype=torch.int32).pin_memory()

    def give(self, t: torch.Tensor) -> None:
        with self._lock:
            self._free.append(t)


_pinned = _PinnedSlots()


_NO_COUNT = 0xFFFFFFFF        # what a pinned word holds until the device has written the instance count


def _await_count(pinned: torch.Tensor, stream) -> int:
    """The asynchronous instance count of a speculative forward.  No event is recorded for it (an event between the two
    phases of the forward costs ~6 us of idle device per frame, tools/trace_gaps.sh): the word starts as _NO_COUNT and is
    written by the counting kernel itself (system-scope store into pinned memory) or by the copy behind it, so the host simply
    looks at it."""
    # busy polling, no sleep: the word normally arrives within ~0.1 ms of the forward starting to execute, and a sleep of
    # "20 us" comes back after 60 us .. 1 ms -- long enough for the device to run dry behind it (25 us of idle in front of every
    # blend_bwd in one of two otherwise identical traces, tools/trace_gaps.sh).  A queue so deep that the count is still
    # missing after ~5 ms of polling is waited for with a stream synchronise instead.
    t_end = time.perf_counter() + 5e-3
    while True:
        n = int(pinned[0].item()) & 0xFFFFFFFF
        if n != _NO_COUNT:
            return n
        if time.perf_counter() > t_end:
            break
    stream.synchronize()                                     # the forward is done: the word must be there
    n = int(pinned[0].item()) & 0xFFFFFFFF
    if n == _NO_COUNT:
        raise RuntimeError("bags_raster: the forward finished without delivering its instance count")
    return n


# Lazy forwards nobody differentiated.  Their finalizer runs wherever the last reference happens to die (inside the cyclic GC,
# inside another forward, at interpreter shutdown), so it must not block, take locks that may be held, or touch the device: it
# only parks the pinned word here.  The next forward (or _drain_abandoned()) looks at the parked words, without waiting.
_abandoned = collections.deque()


def _abandon(pinned, key, capacity):
    _abandoned.append((pinned, key, capacity))               # deque.append is atomic


def _drain_abandoned() -> None:
    for _ in range(len(_abandoned)):
        try:
            pinned, key, capacity = _abandoned.popleft()
        except IndexError:
            return
        n = int(pinned[0].item()) & 0xFFFFFFFF
        if n == _NO_COUNT:                                   # its kernel has not run yet: look again next time
            _abandoned.append((pinned, key, capacity))
            continue
        _pinned.give(pinned)
        _note_count(key, n)
        if n > capacity:
            import warnings
            warnings.warn(f"bags_raster: a lazy forward that was never differentiated held {n} (tile, Gaussian) instances, more "
                          f"than its speculative capacity {capacity}: the image it returned had every tile rendered empty.  "
                          f"Render under torch.no_grad() (such forwards always wait for their count) or leave "
                          f"bags_raster.rasterizer.HOST_WAIT at 'forward'.", RuntimeWarning)


def _resolve(lib, fw: "_Forwarded") -> None:
    """Read the asynchronous instance count of a lazy forward (no-op otherwise).  Overflow: raise, or with LAZY_RECOVER redo
    the second phase on an exact buffer."""
    global LAST_NUM_RENDERED
    if fw.pending is None:
        return
    pinned, fin = fw.pending
    fw.pending = None
    fin.detach()
    n = _await_count(pinned, fw.stream)
    _pinned.give(pinned)
    _note_count(fw.key, n)
    fw.num_rendered = LAST_NUM_RENDERED = n
    if n <= fw.capacity:
        fw.outs = None
        return
    msg = (f"bags_raster: {n} (tile, Gaussian) instances exceeded the speculative capacity {fw.capacity} of a lazy forward "
           f"({CAPACITY_HEADROOM} x the largest count seen for this shape): the image that forward returned had every tile "
           f"rendered empty, so the loss and the cotangent computed from it are wrong.")
    if not LAZY_RECOVER or fw.outs is None:
        fw.outs = None
        raise SpeculationOverflow(msg + "  Redo the step (the capacity hint has been raised), or use the default "
                                        "bags_raster.rasterizer.HOST_WAIT = 'forward'.")
    import warnings
    warnings.warn(msg + "  LAZY_RECOVER: the state is recomputed exactly and the gradients are those of the true render for "
                        "the cotangent that was passed in.", RuntimeWarning)
    pk, dev = fw.packed, fw.packed.device
    color, radii, depth, weights, mean2D = fw.outs
    out = L.BagsForwardOut(color.data_ptr(), radii.data_ptr(), depth.data_ptr(), weights.data_ptr(), mean2D.data_ptr())
    with torch.cuda.device(dev), torch.cuda.stream(fw.stream):
        fw.binning = _bytes(lib.bags_binning_size(n, fw.W, fw.H), dev)
        state = _state_of(fw)
        L.check(lib.bags_forward_finish(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out), n,
                                        fw.stream.cuda_stream), "bags_forward_finish")
    fw.capacity = n
    fw.outs = None


def _finish_wait(lib, fw: "_Forwarded") -> None:
    """Second half of a waiting speculative forward: the host reads the instance count (the device is meanwhile busy with phase 2) and
    redoes phase 2 on an exact buffer if the guess was too small.  No-op for every other kind of forward."""
    global LAST_NUM_RENDERED
    if fw.waiting is None:
        return
    pinned, cap, out = fw.waiting
    fw.waiting = None
    pk, dev = fw.packed, fw.packed.device
    n = _await_count(pinned, fw.stream)
    _pinned.give(pinned)
    _note_count(fw.key, n)
    fw.num_rendered = LAST_NUM_RENDERED = n
    if n <= cap:
        return
    # guess too small (scene changed abruptly): redo the second phase on an exact buffer; phase 1 results stay valid
    with torch.cuda.device(dev), torch.cuda.stream(fw.stream):
        fw.binning = _bytes(lib.bags_binning_size(n, fw.W, fw.H), dev)
        state = _state_of(fw)
        L.check(lib.bags_forward_finish(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out), n, fw.stream.cuda_stream),
                "bags_forward_finish")
    fw.capacity = n


def _run_forward(lib, pk: _Packed, H: int, W: int, speculate: bool = True, lazy: bool = False, prealloc=None, defer_wait: bool = False):
    """prealloc(capacity) -> anything: called once both phases of a waiting speculative forward are enqueued, before the host starts
    to wait for the instance count; its result is kept as fw.pre (PREALLOCATE_BACKWARD).  defer_wait: return before that wait; the
    caller finishes its own bookkeeping first and then calls _finish_wait(lib, fw) -- everything the host does AFTER the count has
    arrived sits between the device's blend_fwd and its blend_bwd."""
    dev, P = pk.device, pk.P
    fw = _Forwarded()
    fw.packed, fw.H, fw.W, fw.pending, fw.outs, fw.pre, fw.waiting = pk, H, W, None, None, None, None
    fw.geom = _bytes(lib.bags_geom_size(P), dev)
    fw.image = _bytes(lib.bags_image_size(W, H), dev)
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    weights = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    mean2D = torch.empty(P, 2, dtype=torch.float32, device=dev)
    out = L.BagsForwardOut(color.data_ptr(), radii.data_ptr(), depth.data_ptr(), weights.data_ptr(), mean2D.data_ptr())
    fw.stream = torch.cuda.current_stream(dev)
    stream = fw.stream.cuda_stream
    state = L.BagsState(fw.geom.data_ptr(), fw.geom.numel(), None, 0, fw.image.data_ptr(), fw.image.numel())
    global LAST_NUM_RENDERED
    fw.key = key = (dev.index, P, W, H)
    if _abandoned:
        _drain_abandoned()
    with _hint_lock:
        hint = _capacity_hint.get(key) if (SPECULATE and speculate and not pk.settings.debug) else None
    lazy = lazy and HOST_WAIT == "lazy"
    if hint is not None:
        cap = _capacity_for(hint, CAPACITY_HEADROOM if lazy else 1.2)
        fw.binning = _bytes(lib.bags_binning_size(cap, W, H), dev)
        state.binning, state.binning_bytes = fw.binning.data_ptr(), fw.binning.numel()
        pinned = _pinned.take()
        pinned[0] = -1                        # _NO_COUNT (the previous user of the slot has read its value)
        L.check(lib.bags_forward_prepare_async(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out),
                                               pinned.data_ptr(), stream), "bags_forward_prepare_async")
        L.check(lib.bags_forward_finish_speculative(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out),
                                                    cap, stream), "bags_forward_finish_speculative")
        fw.capacity = cap
        if lazy:                              # the count is read at backward entry (_resolve)
            fw.num_rendered = None
            # aliases, not the tensors the autograd Function returns: those get a grad_fn that owns ctx, hence fw -- a reference
            # cycle that only the cyclic GC breaks, and until then ~250 MB of state per forward stay allocated (measured: the
            # caching allocator then goes to hipMalloc dozens of times per 40 steps)
            fw.outs = tuple(t.detach() for t in (color, radii, depth, weights, mean2D)) if LAZY_RECOVER else None
            fw.pending = (pinned, weakref.finalize(fw, _abandon, pinned, key, cap))
            return fw, (color, radii, depth, weights, mean2D)
        # phase 2 is already queued behind the count: the device does not wait for us while we wait for the word
        if prealloc is not None:
            fw.pre = prealloc(cap)
        fw.num_rendered = None
        fw.waiting = (pinned, cap, out)
        if not defer_wait:
            _finish_wait(lib, fw)
        return fw, (color, radii, depth, weights, mean2D)
    n = C.c_int64(0)
    L.check(lib.bags_forward_prepare(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out),
                                     C.byref(n), stream), "bags_forward_prepare")
    fw.num_rendered = fw.capacity = int(n.value)
    LAST_NUM_RENDERED = fw.num_rendered
    _note_count(key, fw.num_rendered)
    fw.binning = _bytes(lib.bags_binning_size(fw.num_rendered, W, H), dev)
    state.binning, state.binning_bytes = fw.binning.data_ptr(), fw.binning.numel()
    L.check(lib.bags_forward_finish(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(out),
                                    fw.num_rendered, stream), "bags_forward_finish")
    return fw, (color, radii, depth, weights, mean2D)


def _state_of(fw: _Forwarded) -> L.BagsState:
    return L.BagsState(fw.geom.data_ptr(), fw.geom.numel(), fw.binning.data_ptr(), fw.binning.numel(),
                       fw.image.data_ptr(), fw.image.numel())


def _alloc_backward(lib, need, k, shapes, P, dev, ws_instances, gaussian_grads=True):
    """The tensors a backward writes: the Gaussian-parameter gradients carved out of ONE buffer (64-float aligned slices, so that the
    view-sharded exchange is a single collective over it -- bags_raster/sharding.py finds the common storage; to autograd they are
    ordinary tensors), the per-view gradients, and the record workspace for `ws_instances` instances."""
    def new(shape, flag):
        return torch.empty(shape, dtype=torch.float32, device=dev) if flag else None
    want = _wanted(need, k, shapes, P)
    if not gaussian_grads:                                    # (they accumulate in place: ACCUMULATE_IN_PLACE)
        want = [(n, sh, False) for n, sh, _ in want]
    sizes = {n: (math.prod(sh) if f else 0) for n, sh, f in want}
    total = sum((v + 63) // 64 * 64 for v in sizes.values())
    flat = torch.empty(total, dtype=torch.float32, device=dev) if total else None
    carved, off = {}, 0
    for n, sh, f in want:
        carved[n] = flat[off:off + sizes[n]].view(sh) if f else None
        off += (sizes[n] + 63) // 64 * 64
    del flat
    return dict(carved=carved, means2D=new((P, 3), need[1]), densify=new((P, 3), need[2]), shift=new((3,), need[3]),
                view=new((4, 4), need[10]), proj=new((4, 4), need[11]), intr=new((4, 4), need[12]), campos=new((3,), need[13]),
                ws=_bytes(lib.bags_backward_workspace_size(P, ws_instances), dev), ws_instances=int(ws_instances))


def _wanted(need, k, shapes, P):
    return [("means3D", (P, 3), need[0]),
            ("sh", shapes["sh"], need[4] and k["shs"] is not None),
            ("sh_rest", shapes["sh_rest"], need[15] and k["shs_rest"] is not None),
            ("col", (P, 3), need[5] and k["colors_precomp"] is not None),
            ("opac", shapes["opac"], need[6]),
            ("scales", (P, 3), need[7] and k["scales"] is not None),
            ("rot", (P, 4), need[8] and k["rotations"] is not None),
            ("cov", (P, 6), need[9] and k["cov3D_precomp"] is not None)]


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, means2D_densify, shift_factors, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, viewmatrix, projmatrix, intrinsic, campos, raster_settings, sh_rest=None):
        lib = L.load()
        _require_gpu(means3D)
        with torch.cuda.device(means3D.device):
            pk = _Packed(raster_settings, means3D, means2D, shift_factors, sh, colors_precomp, opacities, scales,
                         rotations, cov3Ds_precomp, viewmatrix, projmatrix, intrinsic, campos, sh_rest)
            shapes = dict(sh=None if sh is None else sh.shape, sh_rest=None if sh_rest is None else sh_rest.shape,
                          opac=opacities.shape, campos=campos.shape)
            need = tuple(ctx.needs_input_grad)
            pre = None
            if PREALLOCATE_BACKWARD and any(need) and not ACCUMULATE_IN_PLACE:
                pre = lambda cap: _alloc_backward(lib, need, pk.keep, shapes, pk.P, pk.device, cap)    # noqa: E731
            fw, outs = _run_forward(lib, pk, int(raster_settings.image_height), int(raster_settings.image_width),
                                    lazy=any(need), prealloc=pre, defer_wait=True)
        ctx.fw = fw
        # (weak: the op must not keep the caller's parameters alive; only used by ACCUMULATE_IN_PLACE)
        ctx.leaves = (tuple(None if t is None else weakref.ref(t) for t in
                            (means3D, sh, sh_rest, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
                      if ACCUMULATE_IN_PLACE else None)
        ctx.shapes = shapes
        color, radii, depth, weights, mean2D = outs
        ctx.mark_non_differentiable(radii, depth, weights, mean2D)
        ctx.set_materialize_grads(False)     # no zero-fill kernels for the four outputs nobody differentiates
        _finish_wait(lib, fw)                # (HOST_WAIT = "forward": the count is read last, when nothing else is left to do here)
        return color, radii, depth, weights, mean2D

    @staticmethod
    def backward(ctx, grad_color, _g_radii, _g_depth, _g_weights, _g_mean2D):
        lib = L.load()
        fw: _Forwarded = ctx.fw
        pk, dev, P = fw.packed, fw.packed.device, fw.packed.P
        need = ctx.needs_input_grad
        if grad_color is None:
            return (None,) * 16
        with torch.cuda.device(dev):
            gc = grad_color.detach()
            if gc.dtype != torch.float32 or not gc.is_contiguous():
                gc = gc.to(torch.float32).contiguous()

            k = pk.keep
            want = _wanted(need, k, ctx.shapes, P)
            # FactoredSH: this view's SH gradient stays factored (12 bytes per Gaussian); FactoredSH.finish forms the rows of the whole step
            fsh = FACTORED_SH if (FACTORED_SH is not None and k["shs"] is not None and (need[4] or need[15])) else None
            # ACCUMULATE_IN_PLACE: every wanted Gaussian gradient has a running sum to be added into
            in_place = None
            if ACCUMULATE_IN_PLACE and ctx.leaves is not None:       # (the flag was already set when this forward ran)
                tgt = {}
                for (n, sh, f), ref in zip(want, ctx.leaves):
                    if not f or (fsh is not None and n in ("sh", "sh_rest")):
                        continue
                    t = ref() if ref is not None else None
                    g = None if t is None else t.grad
                    if (g is None or not t.is_leaf or g.dtype != torch.float32 or not g.is_contiguous() or g.device != dev
                            or tuple(g.shape) != tuple(sh)):
                        tgt = None
                        break
                    tgt[n] = g
                if tgt:
                    in_place = tgt
            # what the forward allocated while it waited for its count (PREALLOCATE_BACKWARD), or the same allocations now.  A lazy
            # forward's workspace is sized for the speculative capacity (>= the count unless the forward overflowed): its count is read
            # below, after the allocations, so that only the struct fill and the call itself sit between the count's arrival and the
            # first backward kernel's launch (tools/trace_gaps.sh: ~40 us of Python there showed up as idle device in front of blend_bwd)
            ws_for = fw.capacity if fw.pending is not None else fw.num_rendered
            pre, fw.pre = fw.pre, None
            if pre is None or pre["ws_instances"] < ws_for:
                pre = _alloc_backward(lib, need, k, ctx.shapes, P, dev, ws_for, gaussian_grads=in_place is None)
            carved = {n: in_place.get(n) for n, _, _ in want} if in_place is not None else pre["carved"]
            g_means3D, g_sh, g_col, g_opac = carved["means3D"], carved["sh"], carved["col"], carved["opac"]
            g_sh_rest = carved["sh_rest"]
            g_dldc = None
            if fsh is not None:
                g_dldc = torch.empty(P, 3, dtype=torch.float32, device=dev)
                # the step's first backward (nothing accumulated yet) lends its carved SH slices to finish(): the finished gradient then
                # sits in the same flat buffer as the other parameters' (GradAllReducer.all_reduce_adopted: ONE collective over it)
                if in_place is None and fsh.target is None:
                    fsh.target = (g_sh, g_sh_rest)
                g_sh = g_sh_rest = None
            if g_dldc is None and k["shs_rest"] is not None and (g_sh is None) != (g_sh_rest is None):
                # the library writes the pair or neither (one staged pass over both): the unwanted half goes to a scratch tensor
                if g_sh is None:
                    g_sh = torch.zeros(ctx.shapes["sh"], dtype=torch.float32, device=dev)
                else:
                    g_sh_rest = torch.zeros(ctx.shapes["sh_rest"], dtype=torch.float32, device=dev)
            g_scales, g_rot, g_cov = carved["scales"], carved["rot"], carved["cov"]
            g_means2D, g_densify, g_shift = pre["means2D"], pre["densify"], pre["shift"]
            g_view, g_proj, g_intr, g_campos = pre["view"], pre["proj"], pre["intr"], pre["campos"]
            ws, ws_for = pre["ws"], pre["ws_instances"]
            del pre
            stream = torch.cuda.current_stream(dev).cuda_stream
            _resolve(lib, fw)
            if fw.num_rendered > ws_for:                         # overflow: the exact redo found more instances than the capacity
                ws = _bytes(lib.bags_backward_workspace_size(P, fw.num_rendered), dev)
            args = L.BagsBackwardArgs(gc.data_ptr(), fw.num_rendered, ws.data_ptr(), ws.numel(), _ptr(g_means3D),
                                      _ptr(g_means2D), _ptr(g_densify), _ptr(g_sh), _ptr(g_col), _ptr(g_opac),
                                      _ptr(g_scales), _ptr(g_rot), _ptr(g_cov), _ptr(g_view), _ptr(g_proj),
                                      _ptr(g_intr), _ptr(g_campos), _ptr(g_shift), fw.capacity, 1 if in_place is not None else 0, int(DENSE_PER_TILE),
                                      _ptr(g_sh_rest))
            args.grad_dldc = _ptr(g_dldc)
            state = _state_of(fw)
            gate = ACCUMULATION_GATE
            if gate is None:
                L.check(lib.bags_backward(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(args), stream),
                        "bags_backward")
            else:
                # views of one step on several streams (AccumulationGate): the per-tile half now, the per-Gaussian half -- the one
                # that adds into the shared gradient buffers -- behind the previous backward's
                ts = torch.cuda.current_stream(dev)
                args.phase = L.BWD_BLEND
                L.check(lib.bags_backward(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(args), stream),
                        "bags_backward (blend half)")
                if gate.event is not None:
                    ts.wait_event(gate.event)
                args.phase = L.BWD_PREPROCESS
                L.check(lib.bags_backward(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), C.byref(args), stream),
                        "bags_backward (per-Gaussian half)")
                gate.event = torch.cuda.Event()
                gate.event.record(ts)
        if g_dldc is not None:                                # (the op's campos tensor is kept alive by the entry: finish reads it)
            fsh.views.append((k["campos"], g_dldc, int(pk.settings.sh_degree), torch.cuda.current_stream(dev)))
            g_sh = g_sh_rest = None
        if g_campos is not None:
            g_campos = g_campos.reshape(ctx.shapes["campos"])
        if in_place is not None:                              # already added into the parameters' .grad: nothing for autograd to add
            g_means3D = g_sh = g_sh_rest = g_col = g_opac = g_scales = g_rot = g_cov = None
        if not need[4]:
            g_sh = None
        if not need[15]:
            g_sh_rest = None
        return (g_means3D, g_means2D, g_densify, g_shift, g_sh, g_col, g_opac, g_scales, g_rot, g_cov, g_view, g_proj,
                g_intr, g_campos, None, g_sh_rest)


def rasterize_gaussians(means3D, means2D, means2D_densify, shift_factors, sh, colors_precomp, opacities, scales,
                        rotations, cov3Ds_precomp, raster_settings: GaussianRasterizationSettings, sh_rest=None):
    return _RasterizeGaussians.apply(means3D, means2D, means2D_densify, shift_factors, sh, colors_precomp, opacities,
                                     scales, rotations, cov3Ds_precomp, raster_settings.viewmatrix,
                                     raster_settings.projmatrix, raster_settings.intrinsic, raster_settings.campos,
                                     raster_settings, sh_rest)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Near-plane frustum test of the stock package (p_view.z > 0.2)."""
        with torch.no_grad():
            v = self.raster_settings.viewmatrix
            z = positions[:, 0] * v[0, 2] + positions[:, 1] * v[1, 2] + positions[:, 2] * v[2, 2] + v[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, means2D_densify=None, shift_factors=None, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, shs_rest=None):
        """The ten keywords of gaussian_renderer/__init__.py:110-121, plus ``shs_rest``: with it, ``shs`` is the reference's
        ``_features_dc`` (P,1,3) and ``shs_rest`` its ``_features_rest`` (P,M-1,3) -- the two parameters as they are stored,
        without ``get_features``' torch.cat in front of the op and the split of dL/dshs behind it."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs_rest is not None and shs is None:
            raise Exception('shs_rest (features_rest) goes with shs = features_dc')
        return rasterize_gaussians(means3D, means2D, means2D_densify, shift_factors, shs, colors_precomp, opacities,
                                   scales, rotations, cov3D_precomp, rs, shs_rest)


def debug_views(settings: GaussianRasterizationSettings, means3D, means2D, shift_factors, shs, colors_precomp,
                opacities, scales, rotations, cov3D_precomp):
    """Forward pass that also returns the integer artefacts (tiles_touched, rect, depth bits, sorted instance list,
    64-bit keys, tile ranges, n_contrib, final_T) for the bit-exact parity tests."""
    lib = L.load()
    _require_gpu(means3D)
    dev = means3D.device
    with torch.cuda.device(dev), torch.no_grad():
        pk = _Packed(settings, means3D, means2D, shift_factors, shs, colors_precomp, opacities, scales, rotations,
                     cov3D_precomp, settings.viewmatrix, settings.projmatrix, settings.intrinsic, settings.campos)
        H, W = int(settings.image_height), int(settings.image_width)
        fw, outs = _run_forward(lib, pk, H, W, speculate=False)
        P, I = pk.P, fw.num_rendered
        T = ((W + 15) // 16) * ((H + 15) // 16)
        i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
        v = dict(tiles_touched=i32(P), rect=i32(P, 4), depth_bits=i32(P), point_list=i32(max(I, 1)),
                 keys_sorted=torch.empty(max(I, 1), dtype=torch.int64, device=dev), ranges=i32(T, 2),
                 n_contrib=i32(H, W), final_T=torch.empty(H, W, dtype=torch.float32, device=dev))
        views = L.BagsDebugViews(*[v[n].data_ptr() for n in ("tiles_touched", "rect", "depth_bits", "point_list",
                                                             "keys_sorted", "ranges", "n_contrib", "final_T")])
        state = _state_of(fw)
        L.check(lib.bags_debug_views(C.byref(pk.settings), C.byref(pk.inputs), C.byref(state), I, C.byref(views),
                                     torch.cuda.current_stream(dev).cuda_stream), "bags_debug_views")
        v["point_list"], v["keys_sorted"] = v["point_list"][:I], v["keys_sorted"][:I]
        v["num_rendered"] = I
        v["outputs"] = outs
    return v


def compute_relocation(opacity_old, scale_old, N, binoms, n_max):
    """utils/reloc_utils.py:11-13.  The reference's only caller is commented out (scene/gaussian_model.py:23,494-504)."""
    raise NotImplementedError("compute_relocation is not part of the accelerated path (MCMC relocation is disabled in the reference)")
