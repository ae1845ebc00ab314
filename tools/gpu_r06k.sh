#!/bin/bash
# round 6: final profile round of the final library (stage events on the dispatches), + the test files that touch the profiler / operator paths
cd "$(dirname "$0")/.."
bash tools/profile_round.sh r06c 2>&1 | tail -4
python - <<'PY'
import json, csv
d=json.load(open('gpurun_out/prof_r06c/bench.json'))
print('bench: ms/step %.4f lazy %.4f median-leg mean %.4f aabb %.4f v4 %.4f'%(d['ms_per_step'], d['config']['other_host_wait']['ms_per_step'], d['median_leg']['mean_ms'], d['config']['aabb']['ms_per_step'], d['v4']['ms_per_view']))
print('stage_ms', d['stage_ms'], 'timed', d['roofline']['launches_timed'])
for r in csv.DictReader(open('gpurun_out/prof_r06c/kernel_stats.csv')):
    if float(r['Percentage'])>0.3: print('rocprof', r['Name'][:36], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py tests/test_flat_grads_gpu.py -x -q -m gpu -k "synthetic or config1 or bitwise or speculative or zero_gaussians or radix or binning_paths_give or backward_twice or several_streams or dense_scene" 2>&1 | tail -3
