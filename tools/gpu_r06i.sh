#!/bin/bash
# round 6: the dominant kernel's events attached to its dispatch -- the timed region against the un-profiled legs, and the event-derived launch
# time against rocprofv3's average of the same library in the same call
cd "$(dirname "$0")/.."
O=gpurun_out/r06i; mkdir -p $O
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-v4-leg --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  median-leg mean %.4f  lazy %.4f  aabb %.4f | blend_bwd %.4f (aabb %.4f)  blend_fwd %.4f'%(d['ms_per_step'], d['median_leg']['mean_ms'], d['config']['other_host_wait']['ms_per_step'], d['config']['aabb']['ms_per_step'], s['blend_bwd'], d['config']['aabb']['roofline']['mean_launch_ms'], s['blend_fwd']))"; }
for rep in 1 2 3; do echo "== default (rep $rep)"; run; echo "== --no-profile (rep $rep)"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-v4-leg --no-aabb-leg --no-lazy-leg --no-median-leg --no-profile --steps 60 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms/step %.4f'%d['ms_per_step'])"; done 2>&1 | tee $O/ab_ext_events_raw.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python3 $OLDPWD/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-60,230-300 | tee -a $O/ab_ext_events_raw.txt
python - <<'PY' | tee -a gpurun_out/r06i/ab_ext_events_raw.txt
import json
d=json.loads(open('gpurun_out/r06i/trace.log').read().strip().splitlines()[-1])
print('under rocprof: ms/step %.4f, event-derived blend_bwd mean %.4f ms'%(d['ms_per_step'], d['stage_ms']['blend_bwd']))
PY
rm -rf $O/trace
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bitwise or synthetic or dense_scene" 2>&1 | tail -3
