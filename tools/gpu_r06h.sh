#!/bin/bash
# round 6: A/B of the backward's chunk geometry rule for the stock tile rule (compacted lists: density of the record holders), + the new
# operator lifecycle test and the aabb parity tests on the new build
cd "$(dirname "$0")/.."
O=gpurun_out/r06h; mkdir -p $O
cp bundle-adjusting-gaussian-splatting_amd/bags_raster/libbags_raster.so tools/ab/r06_tree.so
L="tools/ab/r06_final_before_compact_rule.so tools/ab/r06_tree.so"
for rep in 1 2 3; do LIBS="$L" SMS="0.5" STEPS=60 tools/ab_libs_sweep.sh --tile-bounds aabb; done 2>&1 | tee $O/ab_compact_rule_raw.txt
LIBS="$L" SMS="0.75 1.0 2.0" STEPS=30 tools/ab_libs_sweep.sh --tile-bounds aabb 2>&1 | tee -a $O/ab_compact_rule_raw.txt
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_flat_grads_gpu.py -x -q -m gpu -k "aabb or wide_chunks or backward_twice or binning_paths_agree or tile_bound_modes_render_the_same" > $O/tests.log 2>&1
tail -5 $O/tests.log
