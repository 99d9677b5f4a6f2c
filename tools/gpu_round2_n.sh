#!/bin/bash
mkdir -p gpurun_out/r2n
export TMPDIR=/tmp
SDM_CONV_PC=1 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv3x3_s1 or split or fused or 4gb" > gpurun_out/r2n/ops_pc.log 2>&1
tail -4 gpurun_out/r2n/ops_pc.log
timeout 600 python tools/conv_pc_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n/conv_pc_ab.txt
