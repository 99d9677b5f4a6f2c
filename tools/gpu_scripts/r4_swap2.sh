#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python tools/conv_swap_ab.py > gpurun_out/r4/swap_ab2.txt 2>&1
cat gpurun_out/r4/swap_ab2.txt
