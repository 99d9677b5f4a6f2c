#!/bin/bash
mkdir -p gpurun_out/r2o
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -s -k "fp8_residual" > gpurun_out/r2o/ops_f8.log 2>&1
tail -12 gpurun_out/r2o/ops_f8.log
timeout 600 python tools/conv_pc_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2o/conv_f8_ab.txt
