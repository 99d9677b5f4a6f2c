#!/bin/bash
# whole-step A/B of library variants: bash tools/gpu_scripts/ab_step.sh <tag> <variant.so> ...   ("new" = the product library)
T=$1; shift
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
cp comfyui-sdmatte_amd/csrc/libsdmatte_hip.so /tmp/lib_new.so
for rep in 1 2; do
for m in new "$@"; do
  if [ $m = new ]; then cp /tmp/lib_new.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so; else cp _ab/libsdmatte_hip_$m.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so; fi
  timeout 300 python bench.py --timed-only --steps 4 --warmup 2 > gpurun_out/$T/bench_$m.json 2> gpurun_out/$T/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in list(d["kernel_breakdown_ms"].items())[:5]})
except Exception as e:
    print("$m failed", e)
PY
done
done
cp /tmp/lib_new.so comfyui-sdmatte_amd/csrc/libsdmatte_hip.so
