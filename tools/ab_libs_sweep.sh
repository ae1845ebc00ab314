#!/bin/bash
# GPU box: blend_bwd stage time of several prebuilt libraries over scene densities (same device, one call).
# usage: LIBS="tools/ab/a.so tools/ab/b.so" SMS="0.5 0.75 1.0" tools/ab_libs_sweep.sh [bench args]
cd "$(dirname "$0")/.."
for sm in ${SMS:-0.5 0.75 1.0 2.0}; do
  for lib in $LIBS; do
    BAGS_RASTER_LIB=$(readlink -f $lib) timeout -k 10 200 python bench.py --no-cpu-baseline --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg --steps ${STEPS:-30} --warmup 5 --sm $sm "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('sm $sm  %-28s ms/step %.4f  K1 %.4f emit %.4f blend_fwd %.4f blend_bwd %.4f pre_bwd %.4f  I %s'%('$(basename $lib)', d['ms_per_step'], s.get('preprocess_fwd',0), s.get('tile_sort',0), s['blend_fwd'], s['blend_bwd'], s['preprocess_bwd'], d['config'].get('instances_I')))" || exit 1
  done
done
