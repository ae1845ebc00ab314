#!/bin/bash
# round 6, third GPU call: same-device A/B of four builds (pre-fold, pose fold only, + L2 warming by the waiting waves [the tree], + ids a chunk ahead),
# three rounds alternating at sm 0.5, one at sm 1.0 / 2.0; then the tests that look at the pose gradients and the backward's determinism
cd "$(dirname "$0")/.."
O=gpurun_out/r06c; mkdir -p $O
cp bundle-adjusting-gaussian-splatting_amd/bags_raster/libbags_raster.so tools/ab/r06_tree_warm.so
L="tools/ab/r06_prefold.so tools/ab/r06_fold_only.so tools/ab/r06_tree_warm.so tools/ab/r06_ids_ahead.so"
for rep in 1 2 3; do LIBS="$L" SMS="0.5" STEPS=60 tools/ab_libs_sweep.sh; done 2>&1 | tee $O/ab_fold_warm_raw.txt
LIBS="$L" SMS="1.0 2.0" STEPS=30 tools/ab_libs_sweep.sh 2>&1 | tee -a $O/ab_fold_warm_raw.txt
LIBS="$L" SMS="0.5" STEPS=60 tools/ab_libs_sweep.sh --tile-bounds aabb 2>&1 | tee -a $O/ab_fold_warm_raw.txt
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "synthetic or config1 or bitwise or zero_gaussians or edge_cases or speculative or frozen_camera_mode_against or two_views or extreme or accumulate or split_sh" > $O/tests.log 2>&1
tail -8 $O/tests.log
