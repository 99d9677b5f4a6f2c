#!/bin/bash
# first GPU pass of round 2: denormal probe, precise/fast parity + timing at full size, gpu tests, bench, rocprof
set -x
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/r2a/probe.log 2>&1
import sys, time, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from comfyui_sdmatte_amd import engine as E
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
eng = E.Engine(SDMatteConfig.tiny(), 0, precision="fp16")
# fp16 subnormal operands through the MFMA: w = 3e-6 (subnormal in fp16), x = 1 -> 16*9*3e-6 = 4.3e-4 per output if not flushed
x = torch.ones(1, 16, 16, 16, dtype=torch.float16, device="cuda")
w = torch.full((32, 16, 3, 3), 3e-6)
y = eng.op_conv(x, w, out_f32=True)
print("denorm probe: centre output", float(y[0, 8, 8, 0]), "expected", 16 * 9 * float(torch.tensor(3e-6).half()))
eng.close()
cfg = SDMatteConfig.full()
sd = synthetic_state_dict(cfg, 0)
for prec in ("fp16", "fp16x3"):
    t0 = time.time()
    e = E.Engine(cfg, 0, precision=prec)
    e.load_state_dict(sd)
    print(prec, "load s", time.time() - t0, "blob GB", e.weight_blob_bytes() / 1e9, flush=True)
    for (B, S) in ((1, 512), (4, 1024), (1, 1024)):
        img, tri = synthetic_inputs(B, S, S, seed=1234)
        img, tri = img.cuda(), tri.cuda()
        a = e.apply_matte(img, tri, S)
        ms = []
        for _ in range(3):
            e.apply_matte(img, tri, S); ms.append(e.last_forward_ms())
        print(prec, "B", B, "S", S, "ms", ms, "finite", bool(torch.isfinite(a).all()), flush=True)
    e.profile(True)
    img, tri = synthetic_inputs(4, 1024, 1024, seed=1234)
    e.apply_matte(img.cuda(), tri.cuda(), 1024)
    e.profile(False)
    for k, v in sorted(e.profile_results().items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-14s ms %8.3f n %4d  TF/s %8.1f GB/s %8.1f" % (k, v["ms"], v["launches"], v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6))
    open(f"gpurun_out/r2a/launches_{prec}.csv", "w").write(e.profile_dump())
    e.close()
PY
python -m pytest tests -m gpu -x -q -s > gpurun_out/r2a/gputest.log 2>&1
tail -5 gpurun_out/r2a/gputest.log
python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -c 600 gpurun_out/r2a/bench.json
