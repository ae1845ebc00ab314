#!/bin/bash
# GPU box: per-launch durations of blend_bwd in launch order (clock behaviour over a run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc && rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 $ROOT/bench.py --steps 300 --warmup 5 --settle-steps 0 --no-cpu-baseline --no-profile --no-aabb-leg > /tmp/trc.log 2>&1
f=$(find /tmp/trc -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'blend_bwd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
t0 = int(rows[0]['Start_Timestamp'])
for a, b in ((0, 5), (5, 15), (15, 35), (35, 70), (70, 120), (120, 200), (200, 305)):
    seg = d[a:b]
    if seg: print('launches %3d..%3d  t=%7.1f ms  mean %.1f us  min %.1f  max %.1f' % (a, b, (int(rows[a]['Start_Timestamp']) - t0) / 1e6, sum(seg) / len(seg), min(seg), max(seg)))
PY
