#!/bin/bash
# One call on the GPU box that produces everything profiles/ holds for a version tag:
#   bench JSON (default bench.py run), rocprofv3 --kernel-trace --stats summary of the same workload (without the second,
#   tile_bounds="aabb" leg: it launches the same kernels on a longer list and would be averaged into them), PMC passes
#   (FETCH_SIZE, WRITE_SIZE, SQ instruction / busy counters; separate passes as the guide prescribes).
# usage: tools/profile_round.sh v8        -> gpurun_out/prof_v8/{bench.json,kernel_stats.csv,pmc.txt}
set -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-vX}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done: $(cut -c1-200 $OUT/bench.json)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && echo "kernel stats: $(head -3 $OUT/kernel_stats.csv | cut -c1-160)"
rm -rf $OUT/trace
cd $ROOT
{
  echo "tools/pmc.sh (separate rocprofv3 --pmc passes), $TAG, per-dispatch means"
  tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" blend
  tools/pmc.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" blend
  tools/pmc.sh "FETCH_SIZE" blend
  tools/pmc.sh "WRITE_SIZE" blend
} > $OUT/pmc.txt 2>&1
rm -rf $ROOT/gpurun_out/pmc_*
cat $OUT/pmc.txt
