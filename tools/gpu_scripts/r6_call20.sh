#!/bin/bash
# round 6, call 20: priorities again, one barrier per tile: attn_pp = 1 (static, waves 4-7) vs 5 (softmax segments high + static) vs 4 (softmax segments high only)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
for A in "attn_pp=1" "attn_pp=5" "attn_pp=4" "attn_pp=1" "attn_pp=5"; do
  timeout 600 python bench.py --steps 5 --warmup 2 --timed-only --opt $A > $O/c20_bench_$A.json 2> $O/c20_bench_$A.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6/c20_bench_$A.json').read().strip().splitlines()[-1])
print('$A', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in list(d['kernel_breakdown_ms'].items())[:4]})
PY
done
