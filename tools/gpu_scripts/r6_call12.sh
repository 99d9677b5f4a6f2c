#!/bin/bash
# round 6, call 12: d=512 attention with pipelined fragment reads: op + e2e tests that run it, ablations, step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or d512" > $O/c12_ops.log 2>&1; echo "ops rc=$?" >> $O/c12_ops.log; tail -3 $O/c12_ops.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "d512 or full_model_512_vs" > $O/c12_e2e.log 2>&1; echo "e2e rc=$?" >> $O/c12_e2e.log; grep "max|d|\|passed\|failed\|rc=" $O/c12_e2e.log | cut -c1-200 | tail
timeout 600 python tools/attn_d512_ablate.py > $O/c12_attn_d512_ablate.txt 2>&1; cat $O/c12_attn_d512_ablate.txt
timeout 900 python bench.py --steps 5 --warmup 2 --timed-only > $O/c12_bench.json 2> $O/c12_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c12_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in list(d['kernel_breakdown_ms'].items())[:6]})
PY
