#!/bin/bash
# round 6, call 11: attn_pp default on: attention op tests, e2e subset, bench A/B (prio / no prio / off), SQ counters of the new kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -s -k "attention" > $O/c11_ops.log 2>&1; echo "ops rc=$?" >> $O/c11_ops.log; grep "^\[attn\|passed\|failed\|rc=" $O/c11_ops.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "tiny or full_model_512_vs or full_model_512_batch4 or other_prompt or rect" > $O/c11_e2e.log 2>&1; echo "e2e rc=$?" >> $O/c11_e2e.log; grep "max|d|\|passed\|failed\|rc=" $O/c11_e2e.log | cut -c1-200 | tail -30
for A in "attn_pp=1" "attn_pp=2" "attn_pp=0"; do
  timeout 900 python bench.py --steps 5 --warmup 2 --timed-only --dump-profile $O/c11_per_launch_$A.csv --opt $A > $O/c11_bench_$A.json 2> $O/c11_bench_$A.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6/c11_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('c11_bench_')[1], d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in list(d['kernel_breakdown_ms'].items())[:6]})
    except Exception as ex: print(f, 'ERR', ex)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$O/c11_sq1 -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/$O/c11_sq1.json 2> $R/$O/c11_sq1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL --output-format csv -d $R/$O/c11_sq2 -- python $R/bench.py --timed-only --steps 2 --warmup 1 > $R/$O/c11_sq2.json 2> $R/$O/c11_sq2.err
cd $R
for i in 1 2; do F=$(find $O/c11_sq$i -name "*counter_collection.csv" | head -1); [ -n "$F" ] && python tools/pmc_summary.py $F > $O/c11_sq${i}_by_kernel.csv; done
grep -i "attn_d64\|Kernel_Name" $O/c11_sq1_by_kernel.csv | cut -c1-400
grep -i "attn_d64\|Kernel_Name" $O/c11_sq2_by_kernel.csv | cut -c1-400
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
