#!/bin/bash
mkdir -p gpurun_out/r2s
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv_split or fp8" > gpurun_out/r2s/ops.log 2>&1
tail -3 gpurun_out/r2s/ops.log
timeout 300 python tools/gemm_f8_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s/gemm_f8_ab.txt
timeout 600 python bench.py --no-other-mode > gpurun_out/r2s/bench.json 2> gpurun_out/r2s/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2s/bench.json'))
print(d['value'], d['ms_per_step'], d['parity']['max_abs_dalpha'], d['kernel_breakdown_ms']['gemm_mfma'], d['kernel_breakdown_ms']['conv3x3_mfma'])
PY
