#!/bin/bash
# round-6 evidence on the final tree: PMC traffic + SQ passes, rocprofv3 kernel stats, the default bench line (1024 CPU baseline + parity), stream leg, smoke
T=${1:-r6final}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$PWD
bash tools/gpu_scripts/check_build.sh || exit 9
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/fetch -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/fetch.json 2> $R/gpurun_out/$T/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/write -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/write.json 2> $R/gpurun_out/$T/write.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/sq -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/gpurun_out/$T/sq.json 2> $R/gpurun_out/$T/sq.err
cd $R
F=$(find gpurun_out/$T/fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/$T/write -name "*counter_collection.csv" | head -1); Q=$(find gpurun_out/$T/sq -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W 4 1024 fp16x3 > gpurun_out/$T/pmc_traffic.json
python tools/pmc_summary.py $F > gpurun_out/$T/pmc_fetch_size_by_kernel.csv; python tools/pmc_summary.py $W > gpurun_out/$T/pmc_write_size_by_kernel.csv
python tools/pmc_sq.py $Q 4 1024 fp16x3 gpurun_out/$T/pmc_sq_by_kernel.csv > gpurun_out/$T/pmc_sq.json
cat gpurun_out/$T/pmc_traffic.json gpurun_out/$T/pmc_sq.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_timed -- python $R/bench.py --timed-only > $R/gpurun_out/$T/bench_timed_only.json 2> $R/gpurun_out/$T/stats_timed.err
cd $R
S=$(find gpurun_out/$T/stats_timed -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/$T/rocprofv3_kernel_stats_timed_only.csv
python tools/rocprof_stats_summary.py $S 8 > gpurun_out/$T/family_summary.txt; cat gpurun_out/$T/family_summary.txt
cp gpurun_out/$T/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/$T/pmc_sq.json profiles/pmc_sq.json      # (the bench line below reads them beside the same build)
timeout 900 python bench.py --dump-profile gpurun_out/$T/launches_b4.csv > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
cat gpurun_out/$T/bench.json | cut -c1-4000
timeout 600 python bench.py --stream --steps 3 --warmup 1 > gpurun_out/$T/bench_stream.json 2> gpurun_out/$T/bench_stream.err; cut -c1-700 gpurun_out/$T/bench_stream.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.log 2>&1; tail -2 gpurun_out/$T/smoke.log
find gpurun_out/$T -name "*kernel_trace.csv" -delete
find gpurun_out/$T -name "*counter_collection.csv" -size +8M -delete
find gpurun_out/$T -name "*.db" -delete
