#!/bin/bash
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2p/gputest.log 2>&1
tail -5 gpurun_out/r2p/gputest.log
timeout 600 python bench.py --dump-profile gpurun_out/r2p/launches.csv > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err
cat gpurun_out/r2p/bench.json
