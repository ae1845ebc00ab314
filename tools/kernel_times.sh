#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of a short bench run; prints per-kernel average durations; keeps the CSV as gpurun_out/$1_kernel_stats.csv
TAG=${1:-kt}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg "$@" > $ROOT/gpurun_out/prof_$TAG.log 2>&1
f=$(find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { echo "no stats"; tail -5 $ROOT/gpurun_out/prof_$TAG.log; exit 1; }
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $ROOT/gpurun_out/prof_$TAG
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20:
        us = float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 23.0
        tot += us
        print(r["Name"][:52].ljust(52), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8))
print("sum of per-step kernel time (us):", round(tot, 1))
PY
