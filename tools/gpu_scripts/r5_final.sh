#!/bin/bash
# round 5, final: the full -m gpu suite (with durations), then the evidence passes on the same tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5final
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r5final/pytest_gpu.log 2>&1; tail -20 gpurun_out/r5final/pytest_gpu.log
bash tools/gpu_scripts/r5_evidence.sh r5final
