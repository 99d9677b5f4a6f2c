#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 200 python bench.py --timed-only --steps 8 > gpurun_out/r5/c13_b4.json 2> gpurun_out/r5/c13_b4.err
python -c "import json;d=json.load(open('gpurun_out/r5/c13_b4.json'));print('B=4',d['ms_per_step'],d['value'],{k:v['ms'] for k,v in d['kernel_breakdown_ms'].items()})"
for nw in 0 4 8; do
  timeout 200 python bench.py --timed-only --batch 1 --steps 10 --opt attn_nw=$nw > gpurun_out/r5/c13_b1_nw$nw.json 2> gpurun_out/r5/c13_b1_nw$nw.err
  python -c "import json;d=json.load(open('gpurun_out/r5/c13_b1_nw$nw.json'));print('B=1 attn_nw',$nw,d['ms_per_step'],d['kernel_breakdown_ms']['attn_d64'],d['kernel_breakdown_ms']['gemm_mfma'])"
done
