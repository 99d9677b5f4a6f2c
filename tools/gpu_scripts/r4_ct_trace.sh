#!/bin/bash
# per-launch kernel durations of the VAE encoder with / without the constant-tile path (rocprofv3 kernel trace of two timed steps)
mkdir -p gpurun_out/r4ct
export TMPDIR=/tmp
R=$PWD
cd /tmp
for m in 1 0; do
  rm -rf /tmp/ct_trace_$m
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_trace_$m -- python $R/bench.py --timed-only --steps 1 --warmup 1 --opt trimap_skip=$m > /dev/null 2> $R/gpurun_out/r4ct/trace_$m.err
  T=$(find /tmp/ct_trace_$m -name "*kernel_trace.csv" | head -1)
  python - "$T" > $R/gpurun_out/r4ct/trace_$m.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward of the run: find the last prep_trimap_kernel
idx = max(i for i, r in enumerate(rows) if "prep_trimap" in r["Kernel_Name"])
out = []
for r in rows[idx: idx + 140]:
    n = r["Kernel_Name"]
    short = n[:60]
    out.append(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0:10.1f} us  {short}")
print("\n".join(out))
PY
done
cd $R
