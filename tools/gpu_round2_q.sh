#!/bin/bash
mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fp8_residual or 4gb or split" > gpurun_out/r2q/ops_f8.log 2>&1
tail -3 gpurun_out/r2q/ops_f8.log
timeout 600 python tools/conv_pc_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2q/conv_f8_ab.txt
timeout 600 python bench.py --no-cpu-baseline --no-other-mode > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2q/bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms']['conv3x3_mfma'])
PY
