#!/bin/bash
# round 6: the other configurations and the dense scenes on the final library (one call, one device)
cd "$(dirname "$0")/.."
bash tools/bench_configs.sh r06f 2>&1 | tail -14
for sm in 2.0 3.0; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps 20 --warmup 3 --sm $sm 2>/dev/null | tee -a gpurun_out/configs_r06f.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sm $sm  ms/step %.3f  %.1f M Gaussians/s  I=%d'%(d['ms_per_step'], d['value']/1e6, d['config']['instances_I']), d['stage_ms'])"
done
