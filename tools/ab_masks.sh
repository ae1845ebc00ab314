#!/bin/bash
# GPU box: same-device A/B of the tile masks (D7, second half): stage times and step time with -DNO_TILE_MASKS and without.
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
run() {
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-aabb-leg --steps ${STEPS:-40} --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  I %d  stages:'%(d['ms_per_step'], d['config']['instances_I']), s)"
}
for rep in 1 2; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="-DNO_TILE_MASKS" > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; exit 1; }
  echo "== rectangles only (rep $rep)"; run "$@"
  rm -rf $CS/build && make -C $CS -j8 > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; exit 1; }
  echo "== tile masks (rep $rep)"; run "$@"
done
