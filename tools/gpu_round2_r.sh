#!/bin/bash
mkdir -p gpurun_out/r2r
export TMPDIR=/tmp
for t in 1 2 4 8; do echo "== SDM_CONV_F8_TPB=$t"; SDM_CONV_F8_TPB=$t timeout 300 python tools/conv_pc_ab.py 2>&1 | grep -v amdgpu.ids | grep -E "split " | sed -e 's/4-wave.*| F8/F8/' ; done | tee gpurun_out/r2r/tpb.txt
echo "== auto"; timeout 600 python bench.py --no-cpu-baseline --no-other-mode > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2r/bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms']['conv3x3_mfma'])
PY
