#!/bin/bash
# round 3: where the waves' time goes, per kernel (SQ pass of bench.py --timed-only): parked at s_waitcnt / barrier (WAIT_ANY), issue stalls
# (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY), VALU / LDS / MFMA shares
T=${1:-r3w}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$T/sq2 -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/sq2.json 2> $R/gpurun_out/$T/sq2.err
cd $R
Q=$(find gpurun_out/$T/sq2 -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $Q > gpurun_out/$T/pmc_wave_time_by_kernel.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/$T/pmc_wave_time_by_kernel.csv")))
rows.sort(key=lambda r: -float(r["SQ_BUSY_CYCLES"]))
print("kernel, dispatches, parked(waitcnt/barrier), issue-stalled, issuing, of which VALU, LDS, LDS-issue-stall   (fractions of SQ_WAVE_CYCLES)")
for r in rows[:14]:
    w = float(r["SQ_WAVE_CYCLES"]) or 1.0
    f = lambda k: float(r[k]) / w
    print(f'{r["Kernel_Name"][:78]:78s} {r["Dispatches"]:>5s} {f("SQ_WAIT_ANY"):.3f} {f("SQ_WAIT_INST_ANY"):.3f} {f("SQ_ACTIVE_INST_ANY"):.3f} {f("SQ_ACTIVE_INST_VALU"):.3f} {f("SQ_ACTIVE_INST_LDS"):.3f} {f("SQ_WAIT_INST_LDS"):.3f}')
PY
find gpurun_out/$T -name "*kernel_trace.csv" -delete; find gpurun_out/$T -name "*counter_collection.csv" -delete; find gpurun_out/$T -name "*.db" -delete
