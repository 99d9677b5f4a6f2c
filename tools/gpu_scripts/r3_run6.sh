#!/bin/bash
# round 3, GPU run 6: cross-tile prefetch of the F8 3x3 kernel (SDM_CONV_XTILE) - parity, per-layer A/B, whole step
T=${1:-r3f}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv or gemm or f8" > gpurun_out/$T/ops.log 2>&1; tail -3 gpurun_out/$T/ops.log
timeout 900 python tools/conv_xtile_ab.py > gpurun_out/$T/conv_xtile_ab.txt 2>&1; cat gpurun_out/$T/conv_xtile_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "tiny or full_model_512 or properties" > gpurun_out/$T/e2e.log 2>&1; tail -2 gpurun_out/$T/e2e.log; grep -h "max|d|" gpurun_out/$T/e2e.log
for m in 0 1 0 1; do
  SDM_CONV_XTILE=$m timeout 300 python bench.py --timed-only --steps 4 --warmup 2 > gpurun_out/$T/bench_x$m.json 2> gpurun_out/$T/bench_x$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_x$m.json").read().strip().splitlines()[-1])
    print("xtile$m", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in list(d["kernel_breakdown_ms"].items())[:4]})
except Exception as e:
    print("xtile$m failed", e)
PY
done
