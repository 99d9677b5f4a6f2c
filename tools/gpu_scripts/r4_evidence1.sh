#!/bin/bash
# round 4: full -m gpu suite + the default bench line on the current tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r4/pytest_gpu_full.log 2>&1
tail -5 gpurun_out/r4/pytest_gpu_full.log
tail -8 gpurun_out/r4/pytest_gpu_full.log
