#!/bin/bash
# GPU box: same-device A/B of the whole csrc directory against an alternative copy of it (e.g. the previous commit's, exported
# with `git archive <ref> bundle-adjusting-gaussian-splatting_amd/csrc | tar -x -C tools/ab/prev`).  Builds the alternative into
# the product's place, runs the bench stages, builds the product again, runs them again; REPS times, alternating.
# usage: tools/ab_dir.sh <alternative csrc dir> [bench args]
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
ALT=$1; shift
run() {
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps ${STEPS:-40} --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  cold %.4f  stages:'%(d['ms_per_step'], d['ms_per_step_cold']), s)"
}
rm -rf /tmp/product_csrc && cp -r $CS /tmp/product_csrc
for rep in $(seq 1 ${REPS:-2}); do
  rm -rf $CS && cp -r $ALT $CS && rm -rf $CS/build
  make -C $CS -j8 > /tmp/build.log 2>&1 || { echo "BUILD FAILED (alternative)"; tail -5 /tmp/build.log; rm -rf $CS; cp -r /tmp/product_csrc $CS; exit 1; }
  echo "== alternative (rep $rep)"; run "$@"
  rm -rf $CS && cp -r /tmp/product_csrc $CS && rm -rf $CS/build
  make -C $CS -j8 > /tmp/build.log 2>&1 || { echo "BUILD FAILED (product)"; tail -5 /tmp/build.log; exit 1; }
  echo "== product (rep $rep)"; run "$@"
done
