#!/bin/bash
# round 6, call 1: the plane-fed GEMM on hardware - op parity, per-shape timing, e2e parity at 512^2 full architecture, a short bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_p3" > $O/c1_ops.log 2>&1; echo "ops rc=$?" >> $O/c1_ops.log
timeout 600 python tools/gemm_p3_bench.py --tiles > $O/c1_gemm_bench.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --dump-profile $O/c1_per_launch_b4.csv > $O/c1_bench.json 2> $O/c1_bench.err
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "tiny_core_api or tiny_s192 or full_model_512 or test_e2e_full_model_1024_vs_oracle" > $O/c1_e2e.log 2>&1; echo "e2e rc=$?" >> $O/c1_e2e.log
tail -3 $O/c1_ops.log; cat $O/c1_gemm_bench.txt | tail -30; tail -c 1500 $O/c1_bench.json; tail -5 $O/c1_e2e.log
