#!/bin/bash
# Like sweep_blend.sh for knobs read by any source file: full rebuild per variant.
# usage: tools/sweep_defs.sh "<-D flags of variant 1>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  rm -rf $CS/build
  make -C $CS -j16 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  pre_bwd %.4f pre_fwd %.4f  all:'%(d['ms_per_step'], s['preprocess_bwd'], s['preprocess_fwd']), s)"
done
rm -rf $CS/build; make -C $CS -j16 > /dev/null 2>&1
