#!/bin/bash
# A/B of the per-tile sort's size knobs on two scenes (config 3 and its sm 1.0 variant with ~2.3x longer lists).
# usage: tools/variants_ts.sh "<defs1>" "<defs2>" ...
cd "$(dirname "$0")/.."
CS=bundle-adjusting-gaussian-splatting_amd/csrc
for v in "$@"; do
  rm -rf $CS/build && make -C $CS -j8 DEFS="$v" > /tmp/build.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/build.log; continue; }
  echo "== $v"
  tools/trace_gaps.sh 2>/dev/null | grep -E "tile_sort|per step"
  tools/trace_gaps.sh --sm 1.0 2>/dev/null | grep -E "tile_sort|per step" | sed 's/^/   sm 1.0: /'
done
rm -rf $CS/build; make -C $CS -j8 > /dev/null 2>&1
