#!/bin/bash
# round 6: THE final profile round (library src=432a2a5b05a6 commit=7d518e4) + the default bench a second time
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r06f 2>&1 | tail -3
python bench.py > gpurun_out/prof_r06f/bench_b.json 2> gpurun_out/prof_r06f/bench_b.err
python - <<'PY'
import json, csv
for f in ('bench.json','bench_b.json'):
    d=json.load(open('gpurun_out/prof_r06f/'+f))
    print(f, 'ms/step %.4f = %.1f M/s | lazy %.4f median-leg mean %.4f aabb %.4f v4 %.4f cold %.4f'%(d['ms_per_step'], d['value']/1e6, d['config']['other_host_wait']['ms_per_step'], d['median_leg']['mean_ms'], d['config']['aabb']['ms_per_step'], d['v4']['ms_per_view'], d['ms_per_step_cold']))
    print('   stage_ms', d['stage_ms'], 'timed', d['roofline']['launches_timed'], 'traffic_commit', d['roofline']['traffic_commit'], d['roofline']['library_build'])
    print('   frac', round(d['roofline']['frac'],4), 'valu', round(d['roofline_valu']['valu_issue_frac'],3), 'k1', round(d['roofline_k1']['frac'],3), 'k9', round(d['roofline_k9']['frac'],3), 'op', round(d['op_roofline']['frac'],3), round(d['op_roofline']['frac_wall'],3))
for r in csv.DictReader(open('gpurun_out/prof_r06f/kernel_stats.csv')):
    if float(r['Percentage'])>0.3: print('rocprof', r['Name'][:36], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
