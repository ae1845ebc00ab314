#!/bin/bash
# round 6: the whole GPU suite as the driver runs it, with durations, + smoke
cd "$(dirname "$0")/.."
O=gpurun_out/r06f; mkdir -p $O
( time timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $O/gputest.log 2>&1
tail -40 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
