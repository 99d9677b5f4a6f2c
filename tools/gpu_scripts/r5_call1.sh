#!/bin/bash
# round 5, call 1: the F8 3x3 kernel with ONE A buffer + a ring of four weight steps - op parity on hardware, A/B against the round-4 library, step trace, step time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv or f8" > gpurun_out/r5/c1_ops.txt 2>&1; tail -3 gpurun_out/r5/c1_ops.txt
timeout 600 python tools/conv_libs_ab.py _ab/libsdmatte_r4.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so > gpurun_out/r5/c1_ab.txt 2>&1; cat gpurun_out/r5/c1_ab.txt
SDM_TRACE_LIB=$PWD/tools/_build/libsdmatte_hip_trace.so timeout 600 python tools/conv_trace.py > gpurun_out/r5/c1_trace_new.txt 2>&1; cat gpurun_out/r5/c1_trace_new.txt
SDM_TRACE_LIB=$PWD/_ab/libsdmatte_r4_trace.so timeout 600 python tools/conv_trace.py > gpurun_out/r5/c1_trace_r4.txt 2>&1; grep -A3 "res=1 gn=1 skip=1 blocks 0" gpurun_out/r5/c1_trace_r4.txt | head -30
timeout 300 python bench.py --no-cpu-baseline --no-other-mode --timed-only --steps 8 > gpurun_out/r5/c1_bench.json 2> gpurun_out/r5/c1_bench.err; head -c 700 gpurun_out/r5/c1_bench.json; tail -3 gpurun_out/r5/c1_bench.err
