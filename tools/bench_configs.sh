#!/bin/bash
# The other BASELINE configurations with the current build (one JSON line each): gpurun_out/configs_<tag>.jsonl
cd "$(dirname "$0")/.."
TAG=${1:-vX}
OUT=gpurun_out/configs_$TAG.jsonl
: > $OUT
for args in "--sm 1.0" "--P 2000000" "--P 5000000 --width 3840 --height 2160" "--tile-bounds aabb" "--fixed-pose"; do
  echo "== $args"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $args 2>/dev/null | tee -a $OUT | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step %.3f  %.1f M Gaussians/s  I=%d'%(d['ms_per_step'], d['value']/1e6, d['config']['instances_I']), d['stage_ms'])"
done
echo "== iteration"
timeout -k 10 300 python tools/bench_iteration.py 2>/dev/null | tail -3 | tee gpurun_out/iteration_$TAG.txt
