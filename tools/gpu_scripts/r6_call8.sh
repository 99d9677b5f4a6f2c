#!/bin/bash
# round 6, call 8: ping-pong d=64 attention kernel (attn_pp): op tests, bit-identity + per-shape timings + whole step A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $O/c8_ops.log 2>&1; echo "ops rc=$?" >> $O/c8_ops.log; tail -4 $O/c8_ops.log
timeout 1500 python tools/attn_pp_lab.py > $O/c8_attn_pp_lab.txt 2>&1; cat $O/c8_attn_pp_lab.txt
