#!/bin/bash
# round 4, call 2: spread producer loads + wait-free epilogue: A/B against the round-3 library, timeline, numerics spot check, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 300 python tools/conv_lib_ab.py _ab/libsdmatte_hip_r3.so > gpurun_out/r4/conv_ab2.txt 2>&1
timeout 300 python tools/conv_trace.py > gpurun_out/r4/conv_trace2.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r4/pytest_ops2.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 8 > gpurun_out/r4/bench2.json 2> gpurun_out/r4/bench2.err
cat gpurun_out/r4/conv_ab2.txt; tail -3 gpurun_out/r4/pytest_ops2.log; head -c 300 gpurun_out/r4/bench2.json
