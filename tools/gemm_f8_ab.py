"""A/B of the Linear / 1x1 GEMM kernels on the shapes of the step (random operands): 4-wave split-fp16 kernel (tile cfg 4) vs the
8-wave fp8-residual producer / consumer kernel.  Bench helper.  flag bits: 1 fp32 activations, 2 split precision, 16 fp8 residual terms"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
shapes = [(4, 128, 128, 320, 320), (4, 128, 128, 320, 960), (4, 128, 128, 320, 2560), (4, 128, 128, 1280, 320), (4, 64, 64, 640, 640), (4, 64, 64, 640, 5120),
          (4, 64, 64, 2560, 640), (4, 32, 32, 1280, 1280), (4, 32, 32, 1280, 10240), (4, 32, 32, 5120, 1280), (4, 1024, 1024, 256, 128), (8, 512, 512, 128, 256)]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co
    res = {0: [], 16: []}
    for rep in range(3):
        for f8 in (0, 16):
            res[f8].append(eng.bench_conv(N, H, W, ci, co, ntaps=1, in_f32=3 | f8, tile_cfg=4, iters=8))
    a, b = min(res[0]), min(res[16])
    print(f"M={N * H * W:8d} {ci:5d}->{co:5d}  4-wave {a:7.3f} ms {fl / a / 1e9:6.1f} TF/s | F8 {b:7.3f} ms {fl / b / 1e9:6.1f} TF/s x{a / b:5.3f}", flush=True)
