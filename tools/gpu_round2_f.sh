#!/bin/bash
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
python bench.py --dump-profile gpurun_out/r2f/launches_fp16x3.csv > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
tail -c 1800 gpurun_out/r2f/bench.json
python -m pytest tests -m gpu -q -x > gpurun_out/r2f/gputest.log 2>&1
tail -5 gpurun_out/r2f/gputest.log
