#!/bin/bash
# round 6: FactoredSH (ABI 10) -- parity tests, then the v4 leg with and without it (same device, three rounds), headline beside it
cd "$(dirname "$0")/.."
O=gpurun_out/r06o; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_flat_grads_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "factored or accumulate or several_streams or split_sh or synthetic or bitwise" > $O/tests.log 2>&1; tail -5 $O/tests.log
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-aabb-leg --no-lazy-leg --no-median-leg --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('  ms/step %.4f | v4 ms/view %.4f (factored_sh %s) | pre_bwd %.4f'%(d['ms_per_step'], d['v4']['ms_per_view'], d['v4'].get('factored_sh'), s['preprocess_bwd']))"; }
for rep in 1 2 3; do echo "== factored (rep $rep)"; run; echo "== --no-factored-sh (rep $rep)"; run --no-factored-sh; done 2>&1 | tee $O/ab_factored_sh_raw.txt
