#!/bin/bash
# GPU box: rebuild with -DDIAG_PAIRS, count evaluated / contributing (pixel, splat) pairs on the bench workload, restore the product build.
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
mkdir -p gpurun_out
rm -rf $CS/build && make -C $CS -j8 DEFS="-DDIAG_PAIRS" > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; exit 1; }
timeout 300 python tools/diag_pairs.py > gpurun_out/pairs.json 2> gpurun_out/pairs.err; rc=$?
cat gpurun_out/pairs.json
rm -rf $CS/build && make -C $CS -j8 > /tmp/build.log 2>&1 || { echo "REBUILD FAILED"; exit 1; }
exit $rc
