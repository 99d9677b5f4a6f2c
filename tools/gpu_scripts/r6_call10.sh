#!/bin/bash
# round 6, call 10: LDS-DMA ping-pong attention kernel: op tests, consistency + per-shape timings + whole step A/B, ablations
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $O/c10_ops.log 2>&1; echo "ops rc=$?" >> $O/c10_ops.log; tail -4 $O/c10_ops.log
timeout 600 python tools/attn_pp_ablate.py > $O/c10_attn_pp_ablate.txt 2>&1; cat $O/c10_attn_pp_ablate.txt
timeout 1500 python tools/attn_pp_lab.py > $O/c10_attn_pp_lab.txt 2>&1; grep -v "True$" $O/c10_attn_pp_lab.txt
