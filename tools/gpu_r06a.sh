#!/bin/bash
# round 6, first GPU call: CU-mask probe, the CU-partition experiment, the new full-size parity tests, a 6-rank gloo bench
cd "$(dirname "$0")/.."
O=gpurun_out/r06a; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/cu_mask_probe.hip -o /tmp/cu_mask_probe 2>/dev/null && timeout -k 10 120 /tmp/cu_mask_probe > $O/cu_mask_probe.txt 2>&1
cat $O/cu_mask_probe.txt
timeout -k 10 400 python tools/bench_cu_partition.py > $O/cu_partition.txt 2> $O/cu_partition.err || { echo "cu_partition failed"; tail -5 $O/cu_partition.err; }
grep -v "^{" $O/cu_partition.txt | tail -30
timeout -k 10 300 python bench.py --gpus 6 --backend gloo --P 50000 --width 640 --height 360 --no-cpu-baseline --steps 10 > $O/bench_gloo_6ranks.json 2> $O/bench_gloo_6ranks.err || { echo "gloo bench failed"; tail -5 $O/bench_gloo_6ranks.err; }
cut -c1-400 $O/bench_gloo_6ranks.json
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_robustness_gpu.py -x -q -m gpu --durations=30 \
   -k "config4 or config5 or frozen_camera_mode_at_config2 or render_the_same_at_full_size or config3 or both_modes or test_parity_synthetic or extreme" > $O/tests.log 2>&1
tail -45 $O/tests.log
