#!/bin/bash
# round 3, GPU run 5: cheap knob A/Bs on the whole step (bench.py --timed-only): F8 GEMM threshold, attention block size
T=${1:-r3e}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --timed-only --steps 4 --warmup 2 > gpurun_out/$T/bench_$tag.json 2> gpurun_out/$T/bench_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in list(d["kernel_breakdown_ms"].items())[:5]})
except Exception as e:
    print("$tag failed", e)
PY
}
run base A=1
run gemmf8_640 SDM_GEMM_F8_MIN_K=640
run gemmf8_320 SDM_GEMM_F8_MIN_K=320
run attn_nw4 SDM_ATTN_NW=4
run attn_nw8 SDM_ATTN_NW=8
run tpb2 SDM_CONV_F8_TPB=2
run base2 A=1
