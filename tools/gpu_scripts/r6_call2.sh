#!/bin/bash
# round 6, call 2: what bounds the plane-fed GEMM - tile x stages x ablation lab, then the op tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/gemm_p3_lab.py > $O/c2_lab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_p3" > $O/c2_ops.log 2>&1; echo "ops rc=$?" >> $O/c2_ops.log
cat $O/c2_lab.txt; tail -3 $O/c2_ops.log
