#!/bin/bash
# round 6, call 21: XB (half B issues the DMAs inside its barrier wait): micro A/B + trace, consistency on hardware, step A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6; mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_scripts/check_build.sh || exit 9
timeout 600 python tools/attn_pp_ablate.py > $O/c21_attn_pp_xb.txt 2>&1; grep "trace\]\|ablate=" $O/c21_attn_pp_xb.txt | cut -c1-240
python - > $O/c21_xb_consistency.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
import torch
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16x3")
eng.lib.set_option("attn_pp_min_blocks", 0)
g = torch.Generator(device="cuda").manual_seed(5)
ok = True
for (B, h, Lq, Lk) in ((2, 5, 16384, 16384), (1, 10, 1000, 4160), (2, 20, 256, 16384), (1, 2, 300, 64), (1, 2, 300, 192), (4, 10, 4096, 4096)):
    q = torch.randn(B, Lq, h * 64, generator=g, device="cuda") * 1.5
    k = torch.randn(B, Lk, h * 64, generator=g, device="cuda") * 1.5
    v = torch.randn(B, Lk, h * 64, generator=g, device="cuda")
    bias = torch.where(torch.rand(B, Lk, generator=g, device="cuda") < 0.4, torch.tensor(-10000.0, device="cuda"), torch.tensor(0.0, device="cuda"))
    bias[:, : Lk // 3] = -10000.0
    for bb in (None, bias):
        eng.lib.set_option("attn_pp_xb", 0)
        ref = eng.op_attention_split(q, k, v, h, bias=bb)
        eng.lib.set_option("attn_pp_xb", 1)
        outs = [eng.op_attention_split(q, k, v, h, bias=bb) for _ in range(4)]
        same = all(torch.equal(o, ref) for o in outs)
        ok &= same
        print(B, h, Lq, Lk, "bias" if bb is not None else "dense", "XB == default (4 runs, bit for bit):", same, flush=True)
print("XB consistent:", ok)
PY
tail -3 $O/c21_xb_consistency.txt
for A in "attn_pp_xb=0" "attn_pp_xb=1" "attn_pp_xb=0" "attn_pp_xb=1"; do
  timeout 600 python bench.py --steps 5 --warmup 2 --timed-only --opt $A > $O/c21_bench_$A.json 2> $O/c21_bench_$A.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6/c21_bench_$A.json').read().strip().splitlines()[-1])
print('$A', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in list(d['kernel_breakdown_ms'].items())[:4]})
PY
done
