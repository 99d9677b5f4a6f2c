#!/bin/bash
# round 3, GPU run 4: F8 kernels with [channel][pixel] accumulators and the register-direct epilogue (SDM_CONV_EPI=4)
T=${1:-r3d}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv or gemm or f8" > gpurun_out/$T/ops.log 2>&1; tail -3 gpurun_out/$T/ops.log
timeout 900 python tools/conv_epi_ab.py > gpurun_out/$T/conv_epi_ab.txt 2>&1; cat gpurun_out/$T/conv_epi_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "tiny or full_model_512" > gpurun_out/$T/e2e.log 2>&1; tail -2 gpurun_out/$T/e2e.log; grep -h "max|d|" gpurun_out/$T/e2e.log
for m in 3 4; do
  SDM_CONV_EPI=$m timeout 300 python bench.py --timed-only --steps 4 --warmup 2 --dump-profile gpurun_out/$T/launches_epi$m.csv > gpurun_out/$T/bench_epi$m.json 2> gpurun_out/$T/bench_epi$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_epi$m.json").read().strip().splitlines()[-1])
    print("epi$m", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in d["kernel_breakdown_ms"].items()})
except Exception as e:
    print("epi$m failed", e)
PY
done
