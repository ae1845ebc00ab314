#!/bin/bash
# Like tools/variants.sh, but prints the rocprofv3 kernel durations (tools/trace_gaps.sh) of each -D build: for kernels the
# in-bench stage timers lump together.  usage: tools/variants_kt.sh "<grep pattern>" "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
PAT=$1; shift
for v in "$@"; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  tools/trace_gaps.sh 2>/dev/null | grep -E "$PAT"
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
