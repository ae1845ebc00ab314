#!/bin/bash
# GPU box: the bench at several lengths of the timed region (clock ramp / steady state), one device
cd "$(dirname "$0")/.."
for sw in "30 5 0" "30 5 300" "30 5 100" "300 30 300" "1000 50 0" "30 5 300"; do
  set -- $sw
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --steps $1 --warmup $2 --settle-steps $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('steps %4d warmup %3d settle %4d: ms/step %.4f  bwd %.4f fwd %.4f' % (d['steps'], d['warmup'], d['config']['settle_steps'], d['ms_per_step'], s['blend_bwd'], s['blend_fwd']))"
done
