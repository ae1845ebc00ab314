#!/bin/bash
# rocprofv3 PMC pass over a short bench run; prints per-kernel counter means for the blend kernels.
# usage: tools/pmc.sh "COUNTER1 COUNTER2 ..." [kernel-name-substring]
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_$$
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-profile --no-aabb-leg --no-v4-leg --steps 3 --warmup 1 --settle-steps 20 > /dev/null 2> $OUT/err.log
python3 - "$OUT" "${2:-blend}" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file', open(out + '/err.log').read()[-2000:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if pat in r['Kernel_Name']:
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
