#!/bin/bash
# Runs a list of GPU steps one after the other on the GPU box, each under its own `timeout -k 10`; output of step NAME goes
# to gpurun_out/NAME.log.  A step that times out or is killed ends the call (no further GPU step is started); a step that
# merely fails (test assertion, non-zero exit) does not.
# usage: tools/gpu_steps.sh "name|seconds|command" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
overall=0
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && overall=$rc
done
exit $overall
