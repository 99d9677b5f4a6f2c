#!/bin/bash
# round 6, call 4: attention cores write the GEMM's operand planes; VAE attention linears on the plane-fed GEMM: parity (ops, e2e), bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_p3 or attention" > $O/c4_ops.log 2>&1; echo "ops rc=$?" >> $O/c4_ops.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --dump-profile $O/c4_per_launch_b4.csv > $O/c4_bench.json 2> $O/c4_bench.err
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -k "tiny or full_model_512 or full_model_1024_vs_oracle or d512" > $O/c4_e2e.log 2>&1; echo "e2e rc=$?" >> $O/c4_e2e.log
tail -3 $O/c4_ops.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c4_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('parity'), {k:(v['ms'],v['launches']) for k,v in d['kernel_breakdown_ms'].items()})
PY
tail -5 $O/c4_e2e.log
