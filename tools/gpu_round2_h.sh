#!/bin/bash
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/r2h/gputest.log 2>&1
tail -4 gpurun_out/r2h/gputest.log
python bench.py --dump-profile gpurun_out/r2h/launches_fp16x3.csv > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
tail -c 1500 gpurun_out/r2h/bench.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2h/smoke.log 2>&1; tail -2 gpurun_out/r2h/smoke.log
