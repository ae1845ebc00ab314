#!/bin/bash
# round 6, evidence call on the final library: profile round r06b (rocprof stats + PMC traffic + default bench), the other configurations,
# soak (training-like iterations, operator with PREALLOCATE_BACKWARD), fuzz of the list builders and of the oracle modes
cd "$(dirname "$0")/.."
O=gpurun_out/r06g; mkdir -p $O
bash tools/profile_round.sh r06b 2>&1 | tail -6
bash tools/bench_configs.sh r06b 2>&1 | tail -16
( timeout -k 10 300 python tools/soak.py --iters 400 --P 300000 --check-every 50 2>&1 | tail -2 ) | tee $O/soak_r06.txt
( timeout -k 10 300 python tools/soak.py --iters 300 --P 200000 --tile-bounds aabb --depth-key distance --host-wait lazy 2>&1 | tail -2 ) | tee -a $O/soak_r06.txt
( timeout -k 10 400 python tools/fuzz_paths.py --trials 80 --seed 606 --long --cross-dense 2>&1 | tail -1 ) | tee $O/fuzz_r06_long.json
( timeout -k 10 400 python tools/fuzz_paths.py --trials 40 --seed 607 --oracle 2>&1 | tail -1 ) | tee $O/fuzz_r06_oracle.json
