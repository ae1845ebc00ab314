#!/bin/bash
# round 6, fifth GPU call: (1) A/B of rasterizer.PREALLOCATE_BACKWARD (host side), (2) A/B of FMA contraction in K1's SH block, (3) tests that
# exercise the operator's allocation paths (retain_graph / overflow / accumulate / two streams / frozen camera)
cd "$(dirname "$0")/.."
O=gpurun_out/r06e; mkdir -p $O
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f median %.4f lazy %.4f | K1 %.4f blend_fwd %.4f blend_bwd %.4f pre_bwd %.4f sum %.4f'%(d['ms_per_step'], d.get('ms_per_step_median',0), d['config']['other_host_wait']['ms_per_step'], s.get('preprocess_fwd',0), s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], sum(s.values())))"; }
for rep in 1 2 3; do
  echo "== no prealloc (rep $rep)"; run --no-prealloc
  echo "== prealloc (rep $rep)"; run
  echo "== prealloc + K1 SH fma (rep $rep)"; BAGS_RASTER_LIB=$PWD/tools/ab/r06_k1_sh_fma.so run
done 2>&1 | tee $O/ab_prealloc_k1fma_raw.txt
timeout -k 10 700 python -m pytest tests/test_parity_gpu.py tests/test_flat_grads_gpu.py -x -q -m gpu -k "speculative or bitwise or accumulate or two_views or frozen_camera_mode_against or side_stream or synthetic or zero_gaussians or split_sh or flat or bundle_adjustment" > $O/tests.log 2>&1
tail -6 $O/tests.log
BAGS_RASTER_LIB=$PWD/tools/ab/r06_k1_sh_fma.so timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "synthetic or config1 or extreme or config3_aabb or tile_bound_modes" > $O/tests_k1fma.log 2>&1
tail -6 $O/tests_k1fma.log
