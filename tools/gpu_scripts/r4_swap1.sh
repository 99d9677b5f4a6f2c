#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python tools/conv_swap_ab.py all > gpurun_out/r4/swap_ab1.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r4/pytest_ops_swap1.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 6 > gpurun_out/r4/bench_swap1.json 2> gpurun_out/r4/bench_swap1.err
cat gpurun_out/r4/swap_ab1.txt; tail -4 gpurun_out/r4/pytest_ops_swap1.log; head -c 400 gpurun_out/r4/bench_swap1.json; tail -3 gpurun_out/r4/bench_swap1.err
