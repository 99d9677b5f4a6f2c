#!/bin/bash
# round 3: LDS / issue counters of the F8 3x3 kernel on its two extreme layer classes (is the main loop LDS-bound?)
T=${1:-r3l}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/gpurun_out/$T/sq_counter_names.txt
for shape in "4 1024 1024 128 128" "8 256 256 512 512"; do
  tag=$(echo $shape | tr ' ' '_')
  i=0
  for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/$T/p${i}_$tag -- python $R/tools/pmc_conv.py $shape 247 4 > $R/gpurun_out/$T/p${i}_$tag.out 2> $R/gpurun_out/$T/p${i}_$tag.err
    F=$(find $R/gpurun_out/$T/p${i}_$tag -name "*counter_collection.csv" | head -1)
    if [ -n "$F" ]; then python $R/tools/pmc_summary.py $F | grep -i "conv_mfma\|Kernel" > $R/gpurun_out/$T/p${i}_$tag.csv; cat $R/gpurun_out/$T/p${i}_$tag.csv | cut -c1-700; else tail -3 $R/gpurun_out/$T/p${i}_$tag.err; fi
  done
done
cd $R
find gpurun_out/$T -name "*kernel_trace.csv" -delete; find gpurun_out/$T -name "*counter_collection.csv" -delete; find gpurun_out/$T -name "*.db" -delete
