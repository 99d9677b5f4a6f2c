#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 300 python tools/conv_lib_ab.py _ab/libsdmatte_hip_r3.so > gpurun_out/r4/conv_ab4.txt 2>&1
timeout 300 python tools/conv_trace.py > gpurun_out/r4/conv_trace4.txt 2>&1
timeout 300 python tools/conv_lab.py quick > gpurun_out/r4/conv_lab4.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r4/pytest_ops4.log 2>&1
cat gpurun_out/r4/conv_ab4.txt; tail -3 gpurun_out/r4/pytest_ops4.log; head -16 gpurun_out/r4/conv_trace4.txt
