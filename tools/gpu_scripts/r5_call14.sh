#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > gpurun_out/r5/c14_ops.txt 2>&1; tail -3 gpurun_out/r5/c14_ops.txt
for ks in 1 0; do
  timeout 200 python bench.py --timed-only --batch 1 --steps 10 --opt attn_ksplit=$ks > gpurun_out/r5/c14_b1_ks$ks.json 2> gpurun_out/r5/c14_b1_ks$ks.err
  python -c "import json;d=json.load(open('gpurun_out/r5/c14_b1_ks$ks.json'));print('B=1 attn_ksplit',$ks,d['ms_per_step'],d['kernel_breakdown_ms']['attn_d64'])"
  timeout 200 python bench.py --timed-only --steps 6 --opt attn_ksplit=$ks > gpurun_out/r5/c14_b4_ks$ks.json 2> gpurun_out/r5/c14_b4_ks$ks.err
  python -c "import json;d=json.load(open('gpurun_out/r5/c14_b4_ks$ks.json'));print('B=4 attn_ksplit',$ks,d['ms_per_step'],d['kernel_breakdown_ms']['attn_d64'])"
done
timeout 200 python bench.py --timed-only --batch 2 --steps 8 --opt attn_ksplit=1 > gpurun_out/r5/c14_b2_ks1.json 2>/dev/null; timeout 200 python bench.py --timed-only --batch 2 --steps 8 > gpurun_out/r5/c14_b2_ks0.json 2>/dev/null
python -c "import json;[print('B=2',k,json.load(open('gpurun_out/r5/c14_b2_ks%s.json'%k))['ms_per_step']) for k in '10']"
