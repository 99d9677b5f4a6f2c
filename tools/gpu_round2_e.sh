#!/bin/bash
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
python tools/ablate_conv.py split > gpurun_out/r2e/ablate.log 2>&1
cd /tmp
for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py 4 1024 1024 128 128 3 3 > $GRAFT_REPO_ROOT/gpurun_out/r2e/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/r2e -name "*.csv" | head; du -sh gpurun_out/r2e
grep -v amdgpu.ids gpurun_out/r2e/ablate.log | head -30
