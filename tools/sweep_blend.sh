#!/bin/bash
# Sweeps compile-time knobs that only blend.hip reads (tile -> workgroup mapping, backward row arithmetic): rebuilds that
# one object per variant on the GPU box and prints the bench stage times.
# usage: tools/sweep_blend.sh "<-D flags of variant 1>" "<-D flags of variant 2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  touch $CS/blend.hip
  make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  fwd %.4f bwd %.4f ranges %.4f  all:'%(d['ms_per_step'], s['blend_fwd'], s['blend_bwd'], s['tile_ranges']), s)"
done
touch $CS/blend.hip; make -C $CS -j8 > /dev/null 2>&1
