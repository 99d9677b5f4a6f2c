#!/bin/bash
# round 4, call 1: timeline + differential timing of the F8 3x3 kernel, baseline bench of the tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python tools/conv_trace.py > gpurun_out/r4/conv_trace.txt 2>&1
timeout 600 python tools/conv_lab.py > gpurun_out/r4/conv_lab.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 8 > gpurun_out/r4/bench_base.json 2> gpurun_out/r4/bench_base.err
tail -3 gpurun_out/r4/conv_trace.txt; tail -3 gpurun_out/r4/conv_lab.txt; cat gpurun_out/r4/bench_base.json | head -c 600
