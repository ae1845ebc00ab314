#!/bin/bash
# GPU box: build the library with the given -D flags, run a parity subset and the bench stage times, restore the product build.
# usage: tools/variant_test.sh "<-D flags>" ["<pytest -k expression>"]
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
K=${2:-"synthetic or reproducible or huge or tile_list or saturated or config1"}
rm -rf $CS/build && make -C $CS -j8 DEFS="$1" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $1"; tail -5 /tmp/build.log; exit 1; }
echo "== $1"
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -q -x -k "$K" 2>&1 | tail -3
timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  fwd %.4f bwd %.4f pre_bwd %.4f  all:'%(d['ms_per_step'], s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd']), s)"
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
