"""A/B of two BUILDS of the engine on the F8 conv / GEMM layer shapes that dominate the step (bench helper, not part of the product
path): the product library against another .so (default _ab/libsdmatte_hip_old.so = the previous commit's sources, built in the build
container under _ab/, which is git-ignored but travels to the GPU box), alternating launches in one process so that box-to-box and
clock drift cancel.  usage: python tools/conv_lib_ab.py [other.so]"""
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
shapes = [(4, 1024, 1024, 128, 128, 9), (8, 512, 512, 256, 256, 9), (8, 256, 256, 512, 512, 9), (4, 1024, 1024, 256, 128, 9),
          (4, 128, 128, 320, 320, 9), (4, 64, 64, 640, 640, 9), (4, 32, 32, 1280, 1280, 9),
          (4, 128, 128, 1280, 320, 1), (4, 64, 64, 2560, 640, 1), (4, 32, 32, 5120, 1280, 1)]


def eng_of(path):
    e = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(path)))
    e._on_device = True
    return e


other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "_ab", "libsdmatte_hip_old.so")
new, old = Engine(SDMatteConfig.tiny(), 0, precision="fp16"), eng_of(other)
print(f"A = product library, B = {os.path.relpath(other, ROOT)}")
for (N, H, W, ci, co, nt) in shapes:
    fl = 2.0 * N * H * W * ci * co * nt
    for name, flag in (("conv1 (GN, no res)", BASE | 4), ("conv2 (GN, res)", BASE | 4 | 64)) if nt == 9 else (("gemm (res)", BASE | 64),):
        ta, tb = 1e9, 1e9
        for rep in range(3):
            ta = min(ta, new.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, tile_cfg=0 if nt == 9 else 4, iters=6))
            tb = min(tb, old.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, tile_cfg=0 if nt == 9 else 4, iters=6))
        print(f"N={N} {H}x{W} {ci}->{co} taps={nt} {name:19s} A {ta:7.3f} ms {fl / ta / 1e9:6.1f} TF/s | B {tb:7.3f} ms {fl / tb / 1e9:6.1f} TF/s | x{tb / ta:5.3f}", flush=True)
