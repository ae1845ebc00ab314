#!/bin/bash
# GPU box: the step's kernels in launch order with their mean duration and the mean idle gap in front of each (end of the
# previous kernel on the device -> start of this one), over the steady part of a bench run: what the dependent-launch latency
# of ~12 kernels per step costs.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
# TRACE_ITERATION=1: the same for one training iteration (tools/bench_iteration.py --fused-only: camera chain, activations, render, loss, backward)
if [ -n "$TRACE_ITERATION" ]; then
  rm -rf /tmp/trg && rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $ROOT/tools/bench_iteration.py --fused-only --steps 60 > /tmp/trg.log 2>&1
else
  rm -rf /tmp/trg && rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $ROOT/bench.py --steps 60 --warmup 5 --settle-steps 100 --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg "$@" > /tmp/trg.log 2>&1
fi
f=$(find /tmp/trg -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { echo "no trace"; tail -5 /tmp/trg.log; exit 1; }
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps: from one preprocess_fwd to the next; keep the last 50
starts = [i for i, r in enumerate(rows) if 'preprocess_fwd' in r[2]]
starts = starts[-51:]
dur, gap, cnt, order = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int), []
step_len = []
for a, b in zip(starts[:-1], starts[1:]):
    seq = rows[a:b]
    step_len.append((rows[b][0] - rows[a][0]) / 1e3)
    names = collections.Counter()
    for j, (s, e, n) in enumerate(seq):
        short = n.split('(')[0].replace('void ', '')[:40]
        if short.startswith('at::native'):                     # PyTorch's own: keep the functor, it names the op
            import re, os
            m = re.search(r'(\w+Functor\w*|\w+_kernel_cuda\w*|FillFunctor|\w+Op<\w+>|direct_copy_kernel\w*|\w+_cuda\b)', n)
            short = ('at::' + (m.group(1) if m else n[60:100]))[:40]
        names[short] += 1
        key = (short, names[short])
        if key not in order: order.append(key)
        dur[key] += (e - s) / 1e3; cnt[key] += 1
        prev_end = rows[a + j - 1][1] if (a + j) > 0 else s
        gap[key] += max(0, s - prev_end) / 1e3
tot_d = tot_g = 0.0
for key in order:
    c = cnt[key]
    print('%-42s x%-3d  dur %7.1f us   gap before %5.1f us' % (key[0], c, dur[key] / c, gap[key] / c))
    tot_d += dur[key] / (len(starts) - 1); tot_g += gap[key] / (len(starts) - 1)
print('per step: kernels %.1f us, gaps %.1f us, start-to-start %.1f us over %d steps' % (tot_d, tot_g, sum(step_len) / len(step_len), len(step_len)))
PY
