#!/bin/bash
mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2j/gputest.log 2>&1
tail -3 gpurun_out/r2j/gputest.log
timeout 600 python bench.py > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err
cat gpurun_out/r2j/bench.json
timeout 400 python tools/ablate_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2j/ablate_attn.txt | tail -45
