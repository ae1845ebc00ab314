#!/bin/bash
# round 6: does the headline leg (first timed leg after the settle renders) still run in a clock ramp?  settle 300 / 1500 / 5000 renders, same device
cd "$(dirname "$0")/.."
O=gpurun_out/r06l; mkdir -p $O
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-v4-leg --no-aabb-leg --no-lazy-leg --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stage_ms',{})
print('  ms/step %.4f  median-leg mean %.4f median %.4f | blend_bwd %.4f'%(d['ms_per_step'], d['median_leg']['mean_ms'], d['median_leg']['median_ms'], s.get('blend_bwd',0)))"; }
for rep in 1 2; do
  for n in 300 1500 5000; do echo "== settle $n (rep $rep)"; run --settle-steps $n; done
done 2>&1 | tee $O/ab_settle_raw.txt
