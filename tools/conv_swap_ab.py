"""A/B of the role-swapping F8 3x3 kernel (k_conv_f8s.h, engine option conv_swap = 1) against the one-role-per-wave kernel
(conv_swap = 0, the default) on the layer shapes that dominate the step, in the engine's real I/O format (fp32 in / out, fused GroupNorm + SiLU,
statistics of the consumer, optional fp32 residual; random operands).  Bench helper.  usage: python tools/conv_swap_ab.py [all]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
BASE = 1 | 2 | 16 | 32 | 128                      # fp32 in, split, F8, fp32 out, statistics
shapes = [(4, 1024, 1024, 128, 128), (4, 512, 512, 256, 256), (4, 256, 256, 512, 512), (4, 1024, 1024, 256, 128), (4, 512, 512, 512, 256), (4, 128, 128, 320, 320),
          (4, 64, 64, 640, 640), (4, 32, 32, 1280, 1280), (4, 128, 128, 960, 320), (1, 1024, 1024, 128, 128), (1, 256, 256, 512, 512), (1, 128, 128, 320, 320)]
if len(sys.argv) < 2:
    shapes = shapes[:6]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co * 9
    for name, flag in (("conv1 (GN, no res)", BASE | 4), ("conv2 (GN, res)", BASE | 4 | 64)):
        t = {}
        for rep in range(3):
            for sw in (1, 0):
                eng.lib.set_option("conv_swap", sw)
                t[sw] = min(t.get(sw, 1e9), eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=0, iters=6))
        eng.lib.set_option("conv_swap", 0)
        print(f"N={N} {H}x{W} {ci}->{co} {name:19s} swap {t[1]:7.3f} ms {fl / t[1] / 1e9:6.1f} TF/s | one role per wave {t[0]:7.3f} ms {fl / t[0] / 1e9:6.1f} TF/s | x{t[0] / t[1]:5.3f}", flush=True)
