#!/bin/bash
# round 6, second GPU call: the new full-size parity tests (with durations), then a 4-rank gloo bench (4 ranks + launcher stay inside the pool's
# limit of 6 processes on the card; 6 ranks were killed by the process guard in the first call)
cd "$(dirname "$0")/.."
O=gpurun_out/r06b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py tests/test_robustness_gpu.py -x -q -m gpu --durations=40 \
   -k "config4 or config5 or frozen_camera_mode_at_config2 or render_the_same_at_full_size or config3 or both_modes or test_parity_synthetic or extreme" > $O/tests.log 2>&1
tail -60 $O/tests.log
sleep 2
timeout -k 10 300 python bench.py --gpus 4 --backend gloo --P 50000 --width 640 --height 360 --no-cpu-baseline --steps 10 > $O/bench_gloo_4ranks.json 2> $O/bench_gloo_4ranks.err || { echo "gloo bench failed"; tail -5 $O/bench_gloo_4ranks.err; }
cut -c1-600 $O/bench_gloo_4ranks.json
