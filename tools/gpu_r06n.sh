#!/bin/bash
# round 6, last call: long soak / fuzz of the FINAL library (src=35486adb4d12) and the tests nearest to the last source change
cd "$(dirname "$0")/.."
O=gpurun_out/r06n; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py tests/test_flat_grads_gpu.py tests/test_loss_gpu.py tests/test_camera_gpu.py -x -q -m gpu -k "not full_size and not config1 and not shift_factors and not frustum" > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout -k 10 500 python tools/soak.py --iters 3000 --P 500000 --width 1920 --height 1080 --check-every 250 2>&1 | tail -1 | cut -c1-600 ) | tee $O/soak_long.txt
( timeout -k 10 300 python tools/soak.py --iters 1500 --P 150000 --width 640 --height 360 --tile-bounds aabb --hybrid 2>&1 | tail -1 | cut -c1-400 ) | tee -a $O/soak_long.txt
( timeout -k 10 500 python tools/fuzz_paths.py --trials 200 --seed 9606 --cross-dense 2>&1 | tail -1 ) | tee $O/fuzz_200.json
( timeout -k 10 500 python tools/fuzz_paths.py --trials 60 --seed 9607 --long --cross-dense 2>&1 | tail -1 ) | tee $O/fuzz_long_60.json
