"""A/B of several BUILDS of the engine on the F8 conv layer shapes that dominate the step (bench helper): every library named on the
command line against the first one, launches alternating in one process so that box-to-box and clock drift cancel.
usage: python tools/conv_libs_ab.py base.so other.so [...]"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Bindings, Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
BASE = 1 | 2 | 16 | 32 | 128                      # fp32 in, split, F8, fp32 out, statistics
shapes = [(4, 1024, 1024, 128, 128, 9), (8, 512, 512, 256, 256, 9), (8, 256, 256, 512, 512, 9), (4, 1024, 1024, 256, 128, 9), (4, 128, 128, 320, 320, 9)]
if os.environ.get("SDM_AB_SHAPES") == "all":
    shapes += [(4, 64, 64, 640, 640, 9), (4, 32, 32, 1280, 1280, 9), (4, 128, 128, 1280, 320, 1), (4, 64, 64, 2560, 640, 1)]


def eng_of(path):
    e = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(path)))
    e._on_device = True
    return e


libs = sys.argv[1:]
engs = [eng_of(os.path.join(ROOT, l) if not os.path.isabs(l) else l) for l in libs]
print("libraries: " + " | ".join(f"{i}={os.path.basename(l)}" for i, l in enumerate(libs)))
for (N, H, W, ci, co, nt) in shapes:
    fl = 2.0 * N * H * W * ci * co * nt
    for name, flag in (("conv1", BASE | 4), ("conv2", BASE | 4 | 64)) if nt == 9 else (("gemm", BASE | 64),):
        t = [1e9] * len(engs)
        for rep in range(3):
            for i, e in enumerate(engs):
                t[i] = min(t[i], e.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, tile_cfg=0 if nt == 9 else 4, iters=6))
        print(f"N={N} {H}x{W} {ci}->{co} {name:5s} " + " | ".join(f"{i}: {x:6.3f} ms x{t[0] / x:5.3f}" for i, x in enumerate(t)), flush=True)
