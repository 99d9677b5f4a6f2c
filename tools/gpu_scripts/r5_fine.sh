#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-fine}
export SDM_TRACE_LIB=$PWD/tools/_build/libsdmatte_hip_trace2.so
(timeout 200 python tools/conv_trace_fine.py 4 1024 1024 128 128 1 1; timeout 200 python tools/conv_trace_fine.py 8 256 256 512 512 1 1; timeout 200 python tools/conv_trace_fine.py 4 1024 1024 128 128 0 0) > gpurun_out/r5/${T}.txt 2>&1
cat gpurun_out/r5/${T}.txt
