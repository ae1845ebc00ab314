#!/bin/bash
# round 6: timing events on the dominant kernel's dispatch -- stride 1 / 4 / none, same device; event-derived launch time against rocprofv3
cd "$(dirname "$0")/.."
O=gpurun_out/r06j; mkdir -p $O
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-v4-leg --no-aabb-leg --no-lazy-leg --no-median-leg --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stage_ms',{})
print('  ms/step %.4f  blend_bwd %.4f  timed %s'%(d['ms_per_step'], s.get('blend_bwd',0), d.get('roofline',{}).get('launches_timed')))"; }
for rep in 1 2 3; do
  echo "== stride 1 (rep $rep)"; run --profile-stride 1
  echo "== stride 4 (rep $rep)"; run --profile-stride 4
  echo "== stride 20 (rep $rep)"; run --profile-stride 20
  echo "== no profile (rep $rep)"; run --no-profile
done 2>&1 | tee $O/ab_ext_events_stride_raw.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python3 $OLDPWD/bench.py --steps 100 --warmup 5 --profile-stride 1 --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python - <<'PY' | tee -a gpurun_out/r06j/ab_ext_events_stride_raw.txt
import json, glob, csv
line=[l for l in open('gpurun_out/r06j/trace.log') if l.startswith('{"metric"')][-1]
d=json.loads(line)
f=glob.glob('gpurun_out/r06j/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'blend_bwd' in r['Name']: print('rocprofv3: blend_bwd average %.1f us over %s launches'%(float(r['AverageNs'])/1e3, r['Calls']))
print('same process, events on the dispatch: mean %.1f us over %s launches (ms/step %.4f under the profiler)'%(d['stage_ms']['blend_bwd']*1e3, d['roofline']['launches_timed'], d['ms_per_step']))
PY
rm -rf $O/trace
