#!/bin/bash
# GPU box: A/B of the two binning paths in ONE call (devices differ by several per cent between calls): three alternating runs each.
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for b in radix auto; do
    python bench.py --no-cpu-baseline --no-aabb-leg --steps 30 --binning $b 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$b', 'ms/step %.4f' % d['ms_per_step'], 'fwd %.3f bwd %.3f' % (s['blend_fwd'], s['blend_bwd']), {k: v for k, v in s.items() if k not in ('blend_fwd','blend_bwd')})"
  done
done
