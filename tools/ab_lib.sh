#!/bin/bash
# GPU box: same-device A/B of two BUILDS of the library, both compiled beforehand (the .so files travel with the snapshot; the
# alternative e.g. from `git archive <ref> .../csrc include | tar -x -C /tmp/prev && make -C /tmp/prev/.../csrc OUT=$PWD/tools/ab/x.so`).
# The alternative is loaded through BAGS_RASTER_LIB (bags_raster/_lib.py); REPS alternating runs each.
# usage: tools/ab_lib.sh <alternative .so> [bench args]
cd "$(dirname "$0")/.."
ALT=$(readlink -f $1); shift
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps ${STEPS:-40} --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  K1 %.4f binning %.4f blend_fwd %.4f blend_bwd %.4f pre_bwd %.4f pose_reduce %.4f  I %s'%(d['ms_per_step'], s.get('preprocess_fwd',0), s.get('offsets_scan',0)+s.get('tile_sort',0), s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], s.get('pose_reduce',0), d['config'].get('instances_I')))"
}
for rep in $(seq 1 ${REPS:-2}); do
  echo "== alternative $(basename $ALT) (rep $rep) $@"; BAGS_RASTER_LIB=$ALT run "$@" || exit 1
  echo "== product (rep $rep) $@"; run "$@" || exit 1
done
