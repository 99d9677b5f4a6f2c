#!/bin/bash
# round 3, GPU run 7: where the F8 3x3 kernel's epilogue time goes (finer stamps, stores removed) + store path per CU vs chip
T=${1:-r3g}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
./tools/probe/store_overlap_probe > gpurun_out/$T/store_overlap_probe.txt 2>&1; cat gpurun_out/$T/store_overlap_probe.txt
timeout 300 python tools/conv_trace.py 4 1024 1024 128 128 0 > gpurun_out/$T/trace_128_res0.txt 2>&1; grep -v "block 4" gpurun_out/$T/trace_128_res0.txt
timeout 300 python tools/conv_trace.py 4 1024 1024 128 128 1 > gpurun_out/$T/trace_128_res1.txt 2>&1; grep -v "block 4" gpurun_out/$T/trace_128_res1.txt
