#!/bin/bash
mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -s -k "beyond_4gb or conv3x3_s1" > gpurun_out/r2l/ops.log 2>&1
tail -5 gpurun_out/r2l/ops.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "2048" > gpurun_out/r2l/e2e.log 2>&1
tail -8 gpurun_out/r2l/e2e.log
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
