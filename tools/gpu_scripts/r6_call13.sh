#!/bin/bash
# round 6, call 13: per-launch profile of ONE image at 1024^2 (where does the single-image latency go?)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 600 python bench.py --batch 1 --steps 10 --warmup 3 --timed-only --dump-profile $O/c13_per_launch_b1.csv > $O/c13_bench_b1.json 2> $O/c13_bench_b1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c13_bench_b1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches_per_step'], {k:(v['ms'],v['launches']) for k,v in d['kernel_breakdown_ms'].items()})
PY
