#!/bin/bash
# round 3, GPU run 2: e5m2 activation-side residual operands + per-layer e4m3 weight scales + canonical weight blob on hardware
T=${1:-r3b}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
./tools/probe/f8_semantics_probe > gpurun_out/$T/f8_probe.txt 2>&1; tail -16 gpurun_out/$T/f8_probe.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/$T/ops.log 2>&1; tail -3 gpurun_out/$T/ops.log; grep -h "F8 ranges\|G3 vs" gpurun_out/$T/ops.log
timeout 600 python tools/conv_epi_ab.py quick03 > gpurun_out/$T/conv_epi_ab.txt 2>&1; cat gpurun_out/$T/conv_epi_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "tiny or full_model_512 or weight_blob or node_tail or pre_post" > gpurun_out/$T/e2e.log 2>&1; tail -3 gpurun_out/$T/e2e.log; grep -h "max|d|" gpurun_out/$T/e2e.log
for m in 0 3; do
  SDM_CONV_EPI=$m timeout 300 python bench.py --timed-only --steps 4 --warmup 2 > gpurun_out/$T/bench_epi$m.json 2> gpurun_out/$T/bench_epi$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_epi$m.json").read().strip().splitlines()[-1])
    print("epi$m", d["value"], "img/s", d["ms_per_step"], "ms/step", "load", d["weight_load_s"], {k: v["ms"] for k, v in d["kernel_breakdown_ms"].items()})
except Exception as e:
    print("epi$m failed", e)
PY
done
