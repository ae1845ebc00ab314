#!/bin/bash
# A/B of the loss kernels' -D knobs: rocprofv3 kernel averages of tools/bench_loss.py per build.
# usage: tools/variants_loss.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
ROOT=$PWD
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 $ROOT/tools/bench_loss.py --steps 30 > /tmp/lp.log 2>&1 )
  f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1)
  [ -z "$f" ] && { echo "no stats"; tail -3 /tmp/lp.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("loss_"):
        print("  ", r["Name"][:24].ljust(24), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(10))
PY
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
