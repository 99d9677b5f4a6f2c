#!/bin/bash
# round 3, GPU run 3: fp8 residual terms of Q.K^T (attention PREC = 3) A/B + parity; store/MFMA overlap probe; SQ counter pass
T=${1:-r3c}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$PWD
./tools/probe/store_overlap_probe > gpurun_out/$T/store_overlap_probe.txt 2>&1; cat gpurun_out/$T/store_overlap_probe.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -s -k "attention" > gpurun_out/$T/ops_attn.log 2>&1; tail -2 gpurun_out/$T/ops_attn.log; grep -h "G3 vs\|split attention" gpurun_out/$T/ops_attn.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "tiny or full_model_512 or other_prompt" > gpurun_out/$T/e2e.log 2>&1; tail -2 gpurun_out/$T/e2e.log; grep -h "max|d|" gpurun_out/$T/e2e.log
for m in 0 1; do
  SDM_ATTN_F8=$m timeout 300 python bench.py --timed-only --steps 4 --warmup 2 --dump-profile gpurun_out/$T/launches_attnf8_$m.csv > gpurun_out/$T/bench_attnf8_$m.json 2> gpurun_out/$T/bench_attnf8_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench_attnf8_$m.json").read().strip().splitlines()[-1])
    print("attn_f8=$m", d["value"], "img/s", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in d["kernel_breakdown_ms"].items()})
except Exception as e:
    print("attn_f8=$m failed", e)
PY
done
rocprofv3 -L 2>/dev/null | grep -i "MFMA\|SQ_BUSY_CY\|SQ_WAIT_INST_ANY\|SQ_WAVE_CYCLES\|GRBM_GUI" | head -40 > gpurun_out/$T/counters_avail.txt; head -30 gpurun_out/$T/counters_avail.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/sq -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/sq.json 2> $R/gpurun_out/$T/sq.err
cd $R
C=$(find gpurun_out/$T/sq -name "*counter_collection.csv" | head -1)
if [ -n "$C" ]; then python tools/pmc_sq.py $C 4 1024 fp16x3 gpurun_out/$T/pmc_sq_by_kernel.csv > gpurun_out/$T/pmc_sq.json; cat gpurun_out/$T/pmc_sq.json; head -8 gpurun_out/$T/pmc_sq_by_kernel.csv | cut -c1-250; else tail -5 gpurun_out/$T/sq.err; fi
find gpurun_out/$T -name "*kernel_trace.csv" -delete
find gpurun_out/$T -name "*counter_collection.csv" -size +8M -delete
