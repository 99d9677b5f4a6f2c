"""Differential timing of the F8 3x3 conv kernel (bench helper, -DSDM_CONV_LAB build; results of ablated launches are garbage): which
part of a tile's time goes away when one ingredient is removed.  ConvParams::ablate bits: 1 = activation loads from a cache-resident
64-pixel window, 2 = no weight DMAs after a tile's first two steps, 4 = no MFMAs, 8 = no epilogue stores, 16 = no operand transform /
LDS writes by the producers, 32 = no activation loads after the prologue.
usage: python tools/conv_lab.py [quick]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import ctypes
from comfyui_sdmatte_amd import build as B
from comfyui_sdmatte_amd.engine import Bindings, Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)      # bench-only build, kept out of the package directory (prebuilt in the build container)
lib = B.build_all(extra_flags=("-DSDM_CONV_LAB",), out=os.path.join(ROOT, "tools", "_build", "libsdmatte_hip_lab.so"))
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(lib)))
eng._on_device = True
quick = len(sys.argv) > 1 and sys.argv[1].startswith("quick")
shapes = [(4, 1024, 1024, 128, 128), (8, 512, 512, 256, 256), (8, 256, 256, 512, 512)]
if quick:
    shapes = shapes[:1]
BASE = 1 | 2 | 16 | 32 | 4                      # fp32 in, split, F8, fp32 out, fused GroupNorm + SiLU
VARS = [("full", 0), ("A loads from a resident window", 1), ("no weight DMA", 2), ("no A loads", 32), ("no transform / LDS writes", 16),
        ("no stores", 8), ("no MFMA", 4), ("no MFMA, no stores", 12), ("no MFMA, resident A", 5), ("no MFMA, no DMA", 6),
        ("no MFMA, no DMA, no A loads, no transform, no stores (barrier skeleton)", 4 | 2 | 32 | 16 | 8),
        ("no DMA, no A loads, no transform (consumer alone)", 2 | 32 | 16), ("consumer alone, no stores", 2 | 32 | 16 | 8)]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co * 9
    tiles_cu = N * (H // 8) * (W // 32) * (co // 128) / 256.0
    for name, extra in (("conv1: statistics, no residual", 128), ("conv2: statistics + residual", 128 | 64), ("no statistics, no residual", 0)):
        print(f"== N={N} {H}x{W} {ci}->{co}  {name}  ({tiles_cu:.0f} tiles per CU; MFMA issue per tile {1536 * 6 * (ci // 32)} cycles)", flush=True)
        base = None
        for vn, ab in VARS:
            t = min(eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=BASE | extra, tile_cfg=0, ablate=ab, iters=5) for _ in range(2))
            if base is None:
                base = t
            print(f"   {vn:75s} {t:7.3f} ms  {t * 1e3 / tiles_cu:6.2f} us/tile  {fl / t / 1e9:6.1f} TF/s-equivalent  x{t / base:5.3f}", flush=True)
