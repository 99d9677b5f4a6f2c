#!/bin/bash
# round 6, call 3: blocked operand planes + paired row blocks + rational GELU: lab, op tests, per-shape bench, short bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python tools/gemm_p3_lab.py > $O/c3_lab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_p3" > $O/c3_ops.log 2>&1; echo "ops rc=$?" >> $O/c3_ops.log
timeout 600 python tools/gemm_p3_bench.py > $O/c3_gemm_bench.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --dump-profile $O/c3_per_launch_b4.csv > $O/c3_bench.json 2> $O/c3_bench.err
cat $O/c3_lab.txt; tail -3 $O/c3_ops.log; tail -28 $O/c3_gemm_bench.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c3_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in d['kernel_breakdown_ms'].items()})
PY
