#!/bin/bash
# round 5, call 2: per-step activation loads (one vector per step through asm loads) on top of the ring of four
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
T=${1:-c2}
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv or f8" > gpurun_out/r5/${T}_ops.txt 2>&1; tail -3 gpurun_out/r5/${T}_ops.txt
timeout 600 python tools/conv_libs_ab.py _ab/libsdmatte_r4.so _ab/libsdmatte_v5.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so > gpurun_out/r5/${T}_ab.txt 2>&1; cat gpurun_out/r5/${T}_ab.txt
SDM_TRACE_LIB=$PWD/tools/_build/libsdmatte_hip_trace.so timeout 600 python tools/conv_trace.py > gpurun_out/r5/${T}_trace_new.txt 2>&1; cat gpurun_out/r5/${T}_trace_new.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 8 > gpurun_out/r5/${T}_bench.json 2> gpurun_out/r5/${T}_bench.err; head -c 300 gpurun_out/r5/${T}_bench.json; tail -3 gpurun_out/r5/${T}_bench.err
SDM_TRACE_LIB=$PWD/tools/_build/libsdmatte_hip_trace2.so timeout 200 python tools/conv_trace_fine.py 4 1024 1024 128 128 1 1 > gpurun_out/r5/${T}_fine.txt 2>&1; cat gpurun_out/r5/${T}_fine.txt
