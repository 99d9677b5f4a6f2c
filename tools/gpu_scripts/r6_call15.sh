#!/bin/bash
# round 6, call 15: measured parity values of the full-architecture size / class cases (the suite log runs without -s)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -s -k "node_sizes_640_896_vs_oracle or is_transparent or full_model_512_vs or batch8 or full_model_512_batch4" > $O/c15_parity.log 2>&1; echo "rc=$?" >> $O/c15_parity.log
grep "max|d|\|passed\|failed\|rc=" $O/c15_parity.log | cut -c1-200
timeout 600 python tools/attn_pp_ablate.py > $O/c15_attn_pk_sub.txt 2>&1; cat $O/c15_attn_pk_sub.txt
