"""No-GPU checks of the operator's host logic (bags_raster/rasterizer.py): capacity hints of the speculative forward, the
parking / collection of lazy forwards nobody differentiated, argument validation of the settings added in round 4."""
import warnings

import pytest
import torch

from bags_raster import rasterizer as R


@pytest.fixture(autouse=True)
def _clean_hints():
    saved = (dict(R._capacity_hint), dict(R._below_half), R.HOST_WAIT, R.LAZY_RECOVER)
    R._capacity_hint.clear(); R._below_half.clear(); R._abandoned.clear()
    yield
    R._capacity_hint.clear(); R._capacity_hint.update(saved[0])
    R._below_half.clear(); R._below_half.update(saved[1])
    R.HOST_WAIT, R.LAZY_RECOVER = saved[2], saved[3]
    R._abandoned.clear()


def test_operator_is_exact_by_default():
    """The image a forward returns must always be the true render (train.py:250-331 computes its loss from it): the forward
    reads its instance count before it returns unless the caller opts into the lazy mode, and a lazy overflow raises."""
    assert R.HOST_WAIT == "forward" and R.LAZY_RECOVER is False
    assert issubclass(R.SpeculationOverflow, RuntimeError)


def test_capacity_hint_is_the_maximum_seen_and_comes_down_slowly():
    key = (0, 1000, 64, 48)
    R._note_count(key, 5000)
    R._note_count(key, 3000)                       # smaller counts do not lower it ...
    assert R._capacity_hint[key] == 5000
    R._note_count(key, 7000)
    assert R._capacity_hint[key] == 7000
    for _ in range(255):                           # ... until 256 calls in a row stayed below half of it
        R._note_count(key, 1000)
    assert R._capacity_hint[key] == 7000
    R._note_count(key, 1000)
    assert R._capacity_hint[key] == 2000
    R._note_count(key, 1500)                       # (not below half: the run of small counts starts over)
    assert R._below_half[key] == 0


def test_hint_table_is_a_bounded_lru():
    for p in range(R._HINT_KEYS_MAX + 10):         # densification changes P every few hundred iterations
        R._note_count((0, p, 64, 48), 100 + p)
    assert len(R._capacity_hint) == R._HINT_KEYS_MAX and len(R._below_half) <= R._HINT_KEYS_MAX
    assert (0, 0, 64, 48) not in R._capacity_hint and (0, R._HINT_KEYS_MAX + 9, 64, 48) in R._capacity_hint
    R._note_count((0, 10, 64, 48), 5)              # touching an old key makes it the most recent one
    R._note_count((0, 10_000, 64, 48), 5)
    assert (0, 10, 64, 48) in R._capacity_hint and (0, 11, 64, 48) not in R._capacity_hint


def test_capacities_repeat():
    """Buffer sizes must repeat from call to call (torch's caching allocator): 1/8-octave steps, never below the hint."""
    seen = set()
    for hint in range(100_000, 130_000, 997):
        cap = R._capacity_for(hint, 1.2)
        assert cap >= int(hint * 1.2)
        seen.add(cap)
    assert len(seen) <= 4
    assert R._capacity_for(2_074_322, 4.0) >= 4 * 2_074_322


def test_abandoned_lazy_forward_is_collected_without_blocking_and_reports_an_overflow():
    """The finalizer of a lazy forward nobody differentiated only parks its pinned word (it may run inside the GC, inside
    another forward, at shutdown); the next forward looks at the parked words: a word whose kernel has not run yet stays
    parked, an arrived count feeds the hint, an overflow is reported."""
    key = (0, 77, 64, 48)
    late = torch.tensor([-1], dtype=torch.int32)   # _NO_COUNT: the device has not written it yet
    fits = torch.tensor([900], dtype=torch.int32)
    over = torch.tensor([5000], dtype=torch.int32)
    R._abandon(late, key, 1000); R._abandon(fits, key, 1000); R._abandon(over, key, 1000)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        R._drain_abandoned()
    assert [x for x in w if "never differentiated" in str(x.message)] and len(w) == 1
    assert len(R._abandoned) == 1 and R._abandoned[0][0] is late
    assert R._capacity_hint[key] == 5000
    late[0] = 1200                                 # the count arrives
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        R._drain_abandoned()
    assert not R._abandoned and len(w) == 1        # 1200 > 1000: that forward's image was empty too


def test_settings_reject_unknown_modes_before_touching_the_device():
    from bags_raster import GaussianRasterizationSettings, GaussianRasterizer
    z = torch.zeros
    st = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=z(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), intrinsic=torch.eye(4), sh_degree=0,
                                       campos=z(3))
    assert st.clamp_grad == "stock" and st.tile_bounds == "opacity" and st.binning == "auto"      # the operator's defaults
    with pytest.raises(RuntimeError, match="AMD GPU"):                                             # CPU tensors: no fallback
        GaussianRasterizer(st)(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3), rotations=z(2, 4))



def test_bench_reports_a_failed_start_as_one_json_line():
    """bench.py on a box where it cannot run (here: no GPU) or with a rank count that does not match its launcher must say why on
    STDOUT as one JSON line -- the driver keeps the tail of stdout of a failed SCALE run -- and exit non-zero; it never re-executes
    itself and never runs a smaller bench under the same --gpus."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if __import__("torch").cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU (the error path of a box with one is the rank-count check below)")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    msg = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" in msg and msg["rccl_ranks"] == 0 and msg["n_gpus"] == 1
    # a launcher that started another number of ranks than --gpus says
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2
    msg = json.loads(r.stdout.strip().splitlines()[-1])
    assert "WORLD_SIZE=3" in msg["error"] and msg["n_gpus"] == 2
