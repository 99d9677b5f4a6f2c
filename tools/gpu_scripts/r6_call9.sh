#!/bin/bash
# round 6, call 9: compile-time ablations of the ping-pong attention kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python tools/attn_pp_ablate.py > $O/c9_attn_pp_ablate.txt 2>&1; cat $O/c9_attn_pp_ablate.txt
