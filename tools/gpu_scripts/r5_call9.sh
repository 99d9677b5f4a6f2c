#!/bin/bash
# round 5, call 9: tiles-per-block sweep with the new kernel, single-image per-launch profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
for tpb in 0 2 4 8; do
  timeout 200 python bench.py --timed-only --steps 6 --opt conv_f8_tpb=$tpb > gpurun_out/r5/c9_tpb$tpb.json 2> gpurun_out/r5/c9_tpb$tpb.err
  python -c "import json;d=json.load(open('gpurun_out/r5/c9_tpb$tpb.json'));print('tpb',$tpb,d['ms_per_step'],d['kernel_breakdown_ms']['conv3x3_mfma'])"
done
timeout 300 python bench.py --timed-only --batch 1 --steps 10 --dump-profile gpurun_out/r5/c9_b1_launches.csv > gpurun_out/r5/c9_b1.json 2> gpurun_out/r5/c9_b1.err
python -c "import json;d=json.load(open('gpurun_out/r5/c9_b1.json'));print('B=1',d['ms_per_step']);[print(k,v) for k,v in d['kernel_breakdown_ms'].items()]"
timeout 300 python bench.py --timed-only --steps 6 --dump-profile gpurun_out/r5/c9_b4_launches.csv > gpurun_out/r5/c9_b4.json 2> gpurun_out/r5/c9_b4.err
