#!/bin/bash
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
cat > /tmp/t.py <<'PY'
import sys, os
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
N, H, W, ci, co = [int(x) for x in sys.argv[1:6]]
flag = int(sys.argv[6])
ms = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, ablate=0, iters=5)
print("OK", sys.argv[1:], ms, flush=True)
PY
for pad in 40000 0; do
  for shape in "1 64 64 128 128 3" "1 256 256 128 128 3" "4 1024 1024 128 128 3" "4 1024 1024 128 128 7" "1 64 64 128 3 7"; do
    echo "== pad=$pad shape=$shape" >> gpurun_out/r2d/log.txt
    SDM_SPLIT_LDS_PAD=$pad timeout 120 python /tmp/t.py $shape >> gpurun_out/r2d/log.txt 2>&1
  done
done
grep -v "amdgpu.ids\|^  File\|^Thread\|^$\|Extension" gpurun_out/r2d/log.txt | head -60
