#!/bin/bash
# round 6, call 22: one image: row tile of the plane-fed GEMM (gemm_p3_tile) and the attention threshold
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
for A in "gemm_p3_tile=0" "gemm_p3_tile=128" "gemm_p3_tile=64" "attn_pp_min_blocks=64" "gemm_p3_tile=0"; do
  timeout 600 python bench.py --batch 1 --steps 10 --warmup 3 --timed-only --opt $A > $O/c22_bench_b1_$A.json 2> $O/c22_bench_b1_$A.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6/c22_bench_b1_$A.json').read().strip().splitlines()[-1])
print('$A', d['ms_per_step'], {k:v['ms'] for k,v in list(d['kernel_breakdown_ms'].items())[:5]})
PY
done
