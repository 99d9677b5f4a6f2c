#!/bin/bash
# round 6, call 16: segment trace of the ping-pong attention kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 600 python tools/attn_pp_ablate.py > $O/c16_attn_pp_trace.txt 2>&1; cat $O/c16_attn_pp_trace.txt
