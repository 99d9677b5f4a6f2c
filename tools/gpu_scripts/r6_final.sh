#!/bin/bash
# round 6, final: the full -m gpu suite (with durations), then the evidence passes on the same tree
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r6final}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/$T/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest_gpu.log; tail -20 gpurun_out/$T/pytest_gpu.log
bash tools/gpu_scripts/r6_evidence.sh $T
