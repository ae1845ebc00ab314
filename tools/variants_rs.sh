#!/bin/bash
# GPU box: resample kernel durations (rocprofv3) for several -D builds.  usage: tools/variants_rs.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  tools/kt_script.sh tools/time_resample.py resample
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
