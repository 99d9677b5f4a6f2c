#!/bin/bash
# round 6, call 17: one barrier per tile (OB): op tests, A/B + ablations + trace, lab, step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $O/c17_ops.log 2>&1; echo "ops rc=$?" >> $O/c17_ops.log; tail -3 $O/c17_ops.log
timeout 600 python tools/attn_pp_ablate.py > $O/c17_attn_pp_ob.txt 2>&1; cat $O/c17_attn_pp_ob.txt
timeout 1500 python tools/attn_pp_lab.py > $O/c17_attn_pp_lab.txt 2>&1; grep -v "True$" $O/c17_attn_pp_lab.txt
