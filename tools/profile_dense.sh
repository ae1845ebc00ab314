#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the bench workload at other scene densities (scale multiplier --sm) and on the stock tile rule,
# one summary per case under gpurun_out/prof_dense_<tag>/.   usage: tools/profile_dense.sh r05 "1.0 2.0"
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-vX}
OUT=$ROOT/gpurun_out/prof_dense_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sm in ${2:-1.0 2.0}; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$sm -- python3 $ROOT/bench.py --sm $sm --steps 50 --warmup 5 --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg > $OUT/trace_$sm.log 2>&1 || { echo "sm $sm failed"; tail -3 $OUT/trace_$sm.log; exit 1; }
  f=$(find $OUT/trace_$sm -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_sm$sm.csv && echo "sm $sm: $(tail -1 $OUT/trace_$sm.log | cut -c1-200)" && head -9 $OUT/kernel_stats_sm$sm.csv | cut -c1-140
  rm -rf $OUT/trace_$sm
done
