#!/bin/bash
# Dense scenes (hundreds to thousands of instances per tile), same device: stage times with the backward's dense-scene mode forced
# off (-1: zero records written by blend_bwd's per-tile loops and read by preprocess_bwd), forced on (1: a byte per record) and at
# the library's default threshold (0).  usage: SMS="1.0 1.5 2.0 3.0" tools/ab_dense.sh
cd "$(dirname "$0")/.."
for sm in ${SMS:-1.0 1.5 2.0 3.0}; do
  for thr in ${THRS:--1 1 0}; do
    echo "== sm $sm  dense_per_tile=$thr"
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps 30 --warmup 5 --sm $sm --dense-per-tile $thr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f  K1 %.4f emit %.4f blend_fwd %.4f blend_bwd %.4f pre_bwd %.4f  instances %s'%(d['ms_per_step'], s.get('preprocess_fwd',0), s.get('tile_sort',0), s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], d['config'].get('instances_I')))" || exit 1
  done
done
