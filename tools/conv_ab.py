"""Micro-benchmark of the conv kernel on the layer shapes that dominate the 1024x1024 step (random operands).
Bench helper, not part of the product path.  usage: python tools/conv_ab.py [label]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0)
label = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = [(8, 1024, 1024, 128, 128, 9, 1, 0, (-1,)), (8, 1024, 1024, 128, 128, 9, 1, 1, (-1,)), (8, 512, 512, 256, 256, 9, 1, 0, (-1,)),
          (8, 256, 256, 512, 512, 9, 1, 0, (-1,)), (8, 256, 256, 512, 512, 9, 1, 1, (-1,)), (4, 128, 128, 320, 320, 9, 1, 0, (-1,)),
          (4, 64, 64, 640, 640, 9, 1, 0, (-1,)), (8, 1024, 1024, 128, 128, 9, 2, 1, (-1,)), (4, 1024, 1024, 128, 3, 9, 1, 1, (-1,)),
          (4, 128, 128, 320, 320, 1, 1, 0, (-1,)), (4, 128, 128, 1280, 1280, 1, 1, 0, (-1,)), (4, 1024, 1024, 256, 128, 1, 1, 1, (-1,))]
for (N, H, W, ci, co, nt, st, f32, cfgs) in shapes:
    fl = 2.0 * N * (H // st) * (W // st) * ci * co * nt
    for cfg in cfgs:
        ms = eng.bench_conv(N, H, W, ci, co, ntaps=nt, stride=st, in_f32=f32, tile_cfg=cfg, iters=10)
        print(f"{label:8s} N={N} {H}x{W} {ci}->{co} taps={nt} s={st} f32in={f32} cfg={cfg}: {ms:8.4f} ms {fl / ms / 1e9:8.1f} TF/s")
