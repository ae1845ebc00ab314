#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of any python script of this repo; prints the kernels whose name matches $2.
# usage: tools/kt_script.sh tools/time_resample.py resample
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=$ROOT/$1; PAT=${2:-.}
export TMPDIR=/tmp PYTHONPATH=$ROOT:$ROOT/bundle-adjusting-gaussian-splatting_amd:$ROOT/tests:$PYTHONPATH; cd /tmp
rm -rf /tmp/kts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -- python3 $S > /tmp/kts.log 2>&1
f=$(find /tmp/kts -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { echo "no stats"; tail -5 /tmp/kts.log; exit 1; }
python3 - "$f" "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(r["Name"][:70].ljust(72), r["Calls"].rjust(6), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12))
PY
