#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-c11}
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv or f8 or gemm" > gpurun_out/r5/${T}_ops.txt 2>&1; tail -3 gpurun_out/r5/${T}_ops.txt
SDM_AB_SHAPES=all timeout 600 python tools/conv_libs_ab.py _ab/libsdmatte_r4.so _ab/libsdmatte_v9.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so > gpurun_out/r5/${T}_ab.txt 2>&1; cat gpurun_out/r5/${T}_ab.txt
SDM_TRACE_LIB=$PWD/tools/_build/libsdmatte_hip_trace.so timeout 600 python tools/conv_trace.py > gpurun_out/r5/${T}_trace.txt 2>&1; grep "consumer" gpurun_out/r5/${T}_trace.txt | head -8
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 8 > gpurun_out/r5/${T}_bench.json 2> gpurun_out/r5/${T}_bench.err; head -c 300 gpurun_out/r5/${T}_bench.json
