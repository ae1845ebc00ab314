"""Short forms of the two robustness tools (tools/soak.py, tools/fuzz_paths.py), each in a process of its own so that a device
fault fails the test instead of ending the test run: regimes the parity scenes do not have (depth-clustered long lists, poses
drifting into the scene, splats growing eightfold between two iterations)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=600, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0 and "Memory access fault" not in r.stderr, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
def test_short_soak_with_growing_splats_and_drifting_poses():
    out = _run("soak.py", "--iters", "120", "--P", "120000", "--width", "800", "--height", "448", "--check-every", "25")
    assert out["iters"] == 120 and len(out["losses"]) >= 9 and all(x == x for x in out["losses"])


@pytest.mark.timeout(900)
def test_short_fuzz_of_the_two_list_builders():
    # (--cross-dense: the tile-binned run with a byte per gradient record, the radix run with zero records: still bit-identical)
    out = _run("fuzz_paths.py", "--trials", "30", "--seed", "3", "--long", "--cross-dense")
    assert out["trials"] == 30 and out["failures"] == [], out["failures"]
    # every way through the per-tile sort (csrc/tile_sort.h) was taken by some list of the run: one wave (both sizes), the whole
    # workgroup, depth slabs, the global-memory network, and the LDS bitonic fallback for crowded buckets
    hit = out["sort_paths"]
    for path in ("wave256", "wave512", "block", "slabs", "network"):
        assert any(k.split("+")[0] == path for k in hit), (path, hit)
    assert any(k.endswith("+bitonic") for k in hit), hit


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dense", ["1", "-1"])
def test_both_modes_of_the_backward_match_the_oracle(dense):
    """Dense scenes run the backward with a byte per gradient record (blend_bwd marks what it writes, preprocess_bwd sums only
    that) instead of zero records for the instances behind a tile's deepest contributor (blend.hip, BWD_DENSE_PER_TILE /
    BagsBackwardArgs.dense_per_tile).  The threshold is forced down (1) so that every trial takes that mode, then up (-1) so that
    none does; gradients against the CPU oracle either way."""
    out = _run("fuzz_paths.py", "--trials", "10", "--seed", "11", "--oracle", "--dense", dense)
    assert out["trials"] == 10 and out["failures"] == [], out["failures"]
    # the tool's "second look" (a trial whose only disagreement is a pixel or two on a flipped alpha / transmittance threshold, with
    # the gradient bars widened to what the two oracles differ by themselves) must stay the exception: the seed is pinned, so the
    # number of trials that may land there is too -- a regression between 1e-4 and 3e-4 on several trials fails here
    assert len(out["threshold_pairs"]) <= 1, out["threshold_pairs"]
