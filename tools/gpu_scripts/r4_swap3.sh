#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
SDM_TRACE_RAW=1 timeout 300 python tools/conv_swap_trace.py > gpurun_out/r4/swap_trace3.txt 2>&1
grep -v "^\[trace\]" gpurun_out/r4/swap_trace3.txt | head -30
