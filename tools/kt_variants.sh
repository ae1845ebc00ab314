#!/bin/bash
# GPU box: per-kernel times (rocprofv3) of the bench for several -D builds.  usage: tools/kt_variants.sh "<grep pattern>" "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
PAT=$1; shift
for v in "$@"; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  tools/kernel_times.sh ktv 2>/dev/null | grep -E "$PAT"
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
