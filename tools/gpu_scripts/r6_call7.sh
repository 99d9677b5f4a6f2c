#!/bin/bash
# round 6, call 7: the whole -m gpu suite on the plane-fed GEMM tree (incl. the new full-architecture 640 / 896 / is_transparent / B = 8 cases),
# SQ counter passes with the VALU-side counters for the attention kernels (is d=64 really VALU-co-bound?), default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
R=$PWD
timeout 2700 python -m pytest tests -m gpu -q --durations=15 > $O/c7_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/c7_pytest_gpu.log; tail -25 $O/c7_pytest_gpu.log
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/$O/c7_sq_counter_names.txt
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/$O/c7_sq$i -- python $R/bench.py --timed-only --steps 2 --warmup 1 --no-cpu-baseline --no-other-mode > $R/$O/c7_sq$i.json 2> $R/$O/c7_sq$i.err
  F=$(find $R/$O/c7_sq$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_summary.py $F > $R/$O/c7_sq${i}_by_kernel.csv; else tail -3 $R/$O/c7_sq$i.err; fi
done
cd $R
grep -i "attn_d\|Kernel_Name" $O/c7_sq1_by_kernel.csv | cut -c1-900
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
timeout 900 python bench.py --dump-profile $O/c7_per_launch_b4.csv > $O/c7_bench.json 2> $O/c7_bench.err; cat $O/c7_bench.json | cut -c1-3000
