#!/bin/bash
# A/B of blend_bwd's zero records: per-tile loops (0) against one memset of the record array (threshold in instances per tile).
cd "$(dirname "$0")/.."
for sm in ${SMS:-1.0 2.0 3.0}; do
  for thr in 0 1; do
    echo "== sm $sm  BAGS_PREZERO_PER_TILE=$thr"
    BAGS_PREZERO_PER_TILE=$thr timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps 30 --warmup 5 --sm $sm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  blend_bwd %.4f pre_bwd %.4f  instances %s'%(d['ms_per_step'], s['blend_bwd'], s['preprocess_bwd'], d['config'].get('instances_I')))" || exit 1
  done
done
