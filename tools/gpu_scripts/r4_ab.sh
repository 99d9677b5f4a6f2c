#!/bin/bash
# usage: r4_ab.sh tag lib...   -> gpurun_out/r4/ab_<tag>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
tag=$1; shift
timeout 600 python tools/conv_libs_ab.py "$@" > gpurun_out/r4/ab_$tag.txt 2>&1
cat gpurun_out/r4/ab_$tag.txt
