#!/bin/bash
set -x
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
python tools/ablate_conv.py split > gpurun_out/r2c/ablate.log 2>&1
python bench.py --no-other-mode > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
tail -c 1500 gpurun_out/r2c/bench.json
python -m pytest tests -m gpu -q -s -k "config5 or split_precision or fused_groupnorm" > gpurun_out/r2c/gputest.log 2>&1
tail -5 gpurun_out/r2c/gputest.log
