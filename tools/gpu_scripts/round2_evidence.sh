#!/bin/bash
# round-2 evidence on the final tree: GPU tests, PMC traffic passes, rocprofv3 kernel stats, the bench line
T=${1:-r2z}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$T/gputest.log 2>&1
tail -3 gpurun_out/$T/gputest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/fetch -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/fetch.json 2> $R/gpurun_out/$T/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/write -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/write.json 2> $R/gpurun_out/$T/write.err
cd $R
F=$(find gpurun_out/$T/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/$T/write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W 4 1024 fp16x3 > gpurun_out/$T/pmc_traffic.json && cp gpurun_out/$T/pmc_traffic.json profiles/pmc_traffic.json
cat gpurun_out/$T/pmc_traffic.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -- python $R/bench.py > $R/gpurun_out/$T/bench_under_rocprof.json 2> $R/gpurun_out/$T/stats.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_timed -- python $R/bench.py --timed-only > $R/gpurun_out/$T/bench_timed_only.json 2> $R/gpurun_out/$T/stats_timed.err
cd $R
python tools/rocprof_stats_summary.py $(find gpurun_out/$T/stats_timed -name "*kernel_stats.csv" | head -1) 8 > gpurun_out/$T/family_summary.txt; cat gpurun_out/$T/family_summary.txt
timeout 600 python bench.py --dump-profile gpurun_out/$T/launches.csv > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cat gpurun_out/$T/bench.json
# keep the merged output small
find gpurun_out/$T -name "*kernel_trace.csv" -delete
find gpurun_out/$T -name "*counter_collection.csv" -size +8M -delete
