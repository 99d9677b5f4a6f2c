"""A/B of the Linear / 1x1 GEMM tile configurations on the transformer shapes of the step (split precision, fp32 in / out, residual; random operands).
Bench helper.  cfg 4 = 256x128 (fp32 activations, KC32; the F8 producer / consumer kernel where the layer has the fp8-residual weights),
1 = 128x64, 2 = 64x64.  usage: python tools/gemm_cfg_ab.py [cfg ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
eng._on_device = True
cfgs = [int(x) for x in sys.argv[1:]] or [4, 1, 2]
shapes = [(4, 128, 128, 320, 320), (4, 128, 128, 320, 960), (4, 128, 128, 320, 2560), (4, 128, 128, 1280, 320), (4, 64, 64, 640, 640), (4, 64, 64, 640, 1920),
          (4, 64, 64, 640, 5120), (4, 64, 64, 2560, 640), (4, 32, 32, 1280, 1280), (4, 32, 32, 1280, 3840), (4, 32, 32, 5120, 1280),
          (4, 1024, 1024, 256, 128), (8, 512, 512, 128, 256), (1, 128, 128, 320, 320), (1, 64, 64, 640, 640), (1, 32, 32, 1280, 1280)]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co
    for mname, flag in (("split f32 res", 1 | 2 | 32 | 64), ("split f8 res", 1 | 2 | 16 | 32 | 64)):
        res = {}
        for rep in range(2):
            for cfg in cfgs:
                try:
                    ms = eng.bench_conv(N, H, W, ci, co, ntaps=1, in_f32=flag, tile_cfg=cfg, iters=8)
                except Exception as ex:      # noqa: BLE001
                    ms = float("nan")
                res[cfg] = min(res.get(cfg, 1e9), ms)
        print(f"N={N} {H}x{W} {ci}->{co} {mname:13s} " + "  ".join(f"cfg{c}: {res[c]:7.3f} ms {fl / res[c] / 1e9:6.1f} TF/s" for c in cfgs), flush=True)
