#!/bin/bash
mkdir -p gpurun_out/r2v
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2v/gputest.log 2>&1
tail -2 gpurun_out/r2v/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2v/smoke.log 2>&1; tail -2 gpurun_out/r2v/smoke.log
