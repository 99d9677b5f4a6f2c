"""A/B of the cross-tile prefetch of the F8 3x3 conv kernel (option conv_xtile = 0 / 1) on the layer shapes that dominate the step, in the
engine's real I/O format (fp32 in / out, fused GroupNorm, optional residual, statistics; random operands).  Bench helper."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
shapes = [(8, 1024, 1024, 128, 128), (4, 1024, 1024, 128, 128), (8, 512, 512, 256, 256), (8, 256, 256, 512, 512), (4, 1024, 1024, 256, 128),
          (4, 128, 128, 320, 320), (4, 64, 64, 640, 640), (4, 32, 32, 1280, 1280)]
BASE = 1 | 2 | 16 | 32 | 128 | 4
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co * 9
    for name, flag in (("conv1 (GN, no res)", BASE), ("conv2 (GN, res)", BASE | 64)):
        res = {}
        for rep in range(3):
            for x in ("0", "1"):
                eng.lib.set_option("conv_xtile", int(x))
                t = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=0, iters=6)
                res[x] = min(res.get(x, 1e9), t)
        print(f"N={N} {H}x{W} {ci}->{co} {name:19s} xtile0 {res['0']:7.3f} ms {fl / res['0'] / 1e9:6.1f} TF/s | xtile1 {res['1']:7.3f} ms {fl / res['1'] / 1e9:6.1f} TF/s | x{res['0'] / res['1']:5.3f}",
              flush=True)
eng.lib.set_option("conv_xtile", 1)
print("== tiles per block with the cross-tile prefetch (conv2 form)")
for (N, H, W, ci, co) in shapes[:6]:
    fl = 2.0 * N * H * W * ci * co * 9
    out = []
    for tpb in ("2", "4", "8"):
        eng.lib.set_option("conv_f8_tpb", int(tpb))
        t = min(eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=BASE | 64, tile_cfg=0, iters=6) for _ in range(2))
        out.append(f"tpb{tpb} {t:7.3f} ms {fl / t / 1e9:6.1f}")
    eng.lib.set_option("conv_f8_tpb", 0)
    print(f"N={N} {H}x{W} {ci}->{co}: " + " | ".join(out), flush=True)
