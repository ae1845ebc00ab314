#!/bin/bash
# GPU box: counter groups for the two preprocess kernels, one rocprofv3 --pmc pass each (tools/pmc.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/pmc_pre.txt; : > $OUT
for grp in "$@"; do
  echo "== $grp" >> $OUT
  timeout -k 10 240 bash tools/pmc.sh "$grp" preprocess >> $OUT 2>&1 || exit 1
done
