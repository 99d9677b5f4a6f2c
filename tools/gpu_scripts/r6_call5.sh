#!/bin/bash
# round 6, call 6: persistent tile stream, quad-major fp32 epilogue, 4-rows-per-wave LayerNorm / 16-row GroupNorm-apply plane stores;
# A/B: gemm_p3_persist, gemm_p3_attn; occupancy probe
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_p3 or attention or layernorm or groupnorm" > $O/c6_ops.log 2>&1; echo "ops rc=$?" >> $O/c6_ops.log
python - > $O/c6_occ.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.engine import Engine
eng = Engine(SDMatteConfig.tiny(), 0, True)
eng.lib.set_option("gemm_p3_ablate", 256)
for t in (256, 128, 64):
    eng.lib.set_option("gemm_p3_tile", t)
    for epi in (0, 1, 2, 3, 4):
        print(t, epi, eng.bench_gemm_p3(65536, 320, 2560 if epi == 1 else 320, epi, False, iters=2))
eng.close()
PY
timeout 600 python tools/gemm_p3_bench.py > $O/c6_gemm_bench.txt 2>&1
for A in "gemm_p3_persist=1" "gemm_p3_persist=0" "gemm_p3_attn=0" "gemm_p3=0"; do
  timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-mode --timed-only --opt $A > $O/c6_bench_$A.json 2> $O/c6_bench_$A.err
done
tail -3 $O/c6_ops.log; grep "blocks per CU" $O/c6_occ.txt | sort -u | head -20; tail -27 $O/c6_gemm_bench.txt; python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6/c6_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('c6_bench_')[1], d['value'], d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in d['kernel_breakdown_ms'].items()})
    except Exception as ex: print(f, 'ERR', ex)
PY
