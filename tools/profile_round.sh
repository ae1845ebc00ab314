#!/bin/bash
# One call on the GPU box that produces everything profiles/ holds for a version tag, all from the library that is in the tree:
#   bench JSON (default bench.py run), rocprofv3 --kernel-trace --stats summary of the same workload (without the second,
#   tile_bounds="aabb" leg: it launches the same kernels on a longer list and would be averaged into them), and traffic.json --
#   PMC passes of every kernel (FETCH_SIZE, WRITE_SIZE, SQ instruction / busy / LDS counters; separate passes as the guide
#   prescribes), stamped with the library's build string (tools/make_traffic.py).  bench.py runs LAST once more so that its
#   roofline.traffic comes from the traffic.json just written.
# usage: tools/profile_round.sh r05a        -> gpurun_out/prof_r05a/{bench.json,kernel_stats.csv,traffic.json,traffic_aabb.json}
set -o pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-vX}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
RND=$(echo $TAG | cut -c1-3)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg --no-lazy-leg --no-median-leg > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && echo "kernel stats: $(head -4 $OUT/kernel_stats.csv | cut -c1-160)"
rm -rf $OUT/trace
cd $ROOT
python3 tools/make_traffic.py --out $OUT/traffic.json > $OUT/traffic.log 2>&1 || { echo "make_traffic failed"; tail -5 $OUT/traffic.log; }
tail -2 $OUT/traffic.log
python3 tools/make_traffic.py --out $OUT/traffic_aabb.json --tile-bounds aabb > $OUT/traffic_aabb.log 2>&1 || true
# the bench reads profiles/rNN/traffic.json: put the fresh one there for this run (the caller commits it)
mkdir -p profiles/$RND && cp $OUT/traffic.json profiles/$RND/traffic.json
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done: $(cut -c1-300 $OUT/bench.json)"
