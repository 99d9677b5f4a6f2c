#!/bin/bash
set -x
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
python tools/ablate_conv.py > gpurun_out/r2b/ablate.log 2>&1
tail -3 gpurun_out/r2b/ablate.log
python -m pytest tests -m gpu -q -s > gpurun_out/r2b/gputest.log 2>&1
tail -15 gpurun_out/r2b/gputest.log
