#!/bin/bash
# round-2 evidence: rocprofv3 kernel stats of the default bench command, then FETCH_SIZE / WRITE_SIZE PMC passes (separate runs)
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2k/stats -- python $R/bench.py > $R/gpurun_out/r2k/bench_under_rocprof.json 2> $R/gpurun_out/r2k/stats.err
tail -c 600 $R/gpurun_out/r2k/bench_under_rocprof.json
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2k/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode > $R/gpurun_out/r2k/fetch.json 2> $R/gpurun_out/r2k/fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2k/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode > $R/gpurun_out/r2k/write.json 2> $R/gpurun_out/r2k/write.err
cd $R
find gpurun_out/r2k -name "*.csv" | xargs ls -la
# keep the merged output small: drop the raw kernel traces of the stats pass (only *_stats.csv is kept)
find gpurun_out/r2k/stats -name "*kernel_trace.csv" -delete
