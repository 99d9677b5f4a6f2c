#!/bin/bash
# round 3, GPU run 11: softmax denominator of the split-precision d=64 attention on the matrix pipe (new) vs the VALU sum (old library)
T=${1:-r3k}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -s -k "attention or attn or full_model_512_vs or tiny" > gpurun_out/$T/tests.log 2>&1; tail -3 gpurun_out/$T/tests.log; grep -h "max|d|" gpurun_out/$T/tests.log | head -12
cp comfyui-sdmatte_amd/csrc/libsdmatte_hip.so /tmp/lib_new.so
for m in new old new old; do
  case $m in old) cp _ab/libsdmatte_hip_old.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so;; *) cp /tmp/lib_new.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so;; esac
  timeout 300 python bench.py --timed-only --steps 4 --warmup 2 > gpurun_out/$T/bench_$m.json 2> gpurun_out/$T/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in list(d["kernel_breakdown_ms"].items())[:4]})
except Exception as e:
    print("$m failed", e)
PY
done
cp /tmp/lib_new.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so
