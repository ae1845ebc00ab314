#!/bin/bash
# Sweeps compile-time tuning knobs on the GPU box: rebuilds the library per variant and prints the bench stage times.
# usage: tools/variants.sh "<-D flags of variant 1>" "<-D flags of variant 2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  rm -rf $CS/build
  # a variant of the form "make: VAR=value VAR=value" passes make variables (per-file flags, no blanks inside a value)
  if [[ "$v" == make:* ]]; then read -ra MA <<< "${v#make:}"; else MA=("DEFS=$v"); fi
  make -C $CS -j8 "${MA[@]}" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  timeout 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps ${STEPS:-40} --warmup 5 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  fwd %.4f bwd %.4f pre_bwd %.4f binning %.4f  all:'%(d['ms_per_step'], s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], sum(v for k, v in s.items() if k in ('depth_sort','offsets_scan','emit','tile_sort','tile_ranges'))), s)"
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
