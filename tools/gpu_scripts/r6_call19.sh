#!/bin/bash
# round 6, call 19: threshold for the ping-pong kernel: attention op tests, tiny e2e through each kernel, step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $O/c19_ops.log 2>&1; echo "ops rc=$?" >> $O/c19_ops.log; tail -3 $O/c19_ops.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "each_attention_pipeline_kernel or full_model_512_vs" > $O/c19_e2e.log 2>&1; echo "e2e rc=$?" >> $O/c19_e2e.log; grep "max|d|\|passed\|failed\|rc=" $O/c19_e2e.log | cut -c1-200
for A in "attn_pp_min_blocks=128" "attn_pp_min_blocks=0" "attn_pp_min_blocks=128" "attn_pp_min_blocks=0"; do
  timeout 600 python bench.py --steps 5 --warmup 2 --timed-only --opt $A > $O/c19_bench_$A.json 2> $O/c19_bench_$A.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6/c19_bench_$A.json').read().strip().splitlines()[-1])
print('$A', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in list(d['kernel_breakdown_ms'].items())[:4]})
PY
done
