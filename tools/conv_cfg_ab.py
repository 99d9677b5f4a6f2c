"""A/B of conv tile configurations on the layer shapes that dominate the step (random operands).  Bench helper.
flag bits: 1 fp32 activations, 2 split precision, 4 fused GroupNorm+SiLU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
cfgs = [int(x) for x in sys.argv[1:]] or [0, 5]
shapes = [(4, 1024, 1024, 128, 128), (8, 1024, 1024, 128, 128), (4, 512, 512, 256, 256), (4, 256, 256, 512, 512), (4, 128, 128, 512, 512),
          (4, 128, 128, 320, 320), (4, 64, 64, 640, 640), (4, 32, 32, 1280, 1280)]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co * 9
    for mname, flag in (("split+GN", 7), ("fp32+GN", 5), ("fp16", 0)):
        res = []
        for rep in range(2):
            for cfg in cfgs:
                ms = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=cfg, iters=8)
                res.append((cfg, ms))
        best = {c: min(m for cc, m in res if cc == c) for c in cfgs}
        print(f"N={N} {H}x{W} {ci}->{co} {mname:9s} " + "  ".join(f"cfg{c}: {best[c]:7.3f} ms {fl / best[c] / 1e9:7.1f} TF/s" for c in cfgs), flush=True)
