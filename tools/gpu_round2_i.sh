#!/bin/bash
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" > gpurun_out/r2i/ops.log 2>&1
tail -4 gpurun_out/r2i/ops.log
echo "== DMA on"; timeout 300 python tools/conv_cfg_ab.py 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2i/ab_dma1.log
echo "== DMA off"; SDM_CONV_DMA=0 timeout 300 python tools/conv_cfg_ab.py 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2i/ab_dma0.log
