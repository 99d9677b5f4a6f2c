#!/bin/bash
# round 3: the whole -m gpu suite on the final tree (what the driver runs at round end)
T=${1:-r3t}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -s --durations=15 > gpurun_out/$T/gputest.log 2>&1
tail -30 gpurun_out/$T/gputest.log
grep -h "max|d|" gpurun_out/$T/gputest.log | head -60
