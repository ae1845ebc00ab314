#!/bin/bash
# round 6, fourth GPU call: (1) HIP_FORCE_DEV_KERNARG A/B (kernel arguments in device memory instead of host-coherent memory: shorter dispatch
# of every launch), three alternating rounds; (2) tools/profile_round.sh r06a: rocprof stats + PMC traffic of the library in the tree + bench
cd "$(dirname "$0")/.."
O=gpurun_out/r06d; mkdir -p $O
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --steps 60 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f median %.4f  K1 %.4f prefix %.4f emit %.4f blend_fwd %.4f blend_bwd %.4f pre_bwd %.4f pose_reduce %.4f sum %.4f'%(d['ms_per_step'], d.get('ms_per_step_median',0), s.get('preprocess_fwd',0), s.get('offsets_scan',0), s.get('tile_sort',0), s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], s.get('pose_reduce',0), sum(s.values())))"; }
for rep in 1 2 3; do
  echo "== HIP_FORCE_DEV_KERNARG=0 (rep $rep)"; HIP_FORCE_DEV_KERNARG=0 run
  echo "== HIP_FORCE_DEV_KERNARG=1 (rep $rep)"; HIP_FORCE_DEV_KERNARG=1 run
  echo "== unset (rep $rep)"; run
done 2>&1 | tee $O/ab_kernarg_raw.txt
bash tools/profile_round.sh r06a 2>&1 | tail -12
