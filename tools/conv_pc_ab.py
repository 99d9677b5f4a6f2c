"""A/B of the producer / consumer form (8 waves, 1 block per CU) against the 4-wave form (2 blocks per CU) of the split-precision
DMA-weight conv3x3 kernel on the layer shapes that dominate the step (random operands).  Bench helper.
flag bits: 1 fp32 activations, 2 split precision, 4 fused GroupNorm+SiLU, 8 producer / consumer form, 16 fp8 residual terms"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
shapes = [(4, 1024, 1024, 128, 128), (8, 1024, 1024, 128, 128), (4, 512, 512, 256, 256), (8, 512, 512, 256, 256), (4, 256, 256, 512, 512),
          (4, 128, 128, 512, 512), (4, 1024, 1024, 256, 128), (4, 128, 128, 320, 320), (4, 64, 64, 640, 640), (4, 32, 32, 1280, 1280)]
for (N, H, W, ci, co) in shapes:
    fl = 2.0 * N * H * W * ci * co * 9
    for mname, flag in (("split+GN", 7), ("split", 3)):
        res = {0: [], 8: [], 16: []}
        for rep in range(3):
            for pc in (0, 8, 16):
                res[pc].append(eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag | pc, tile_cfg=0, iters=8))
        a, b, c = min(res[0]), min(res[8]), min(res[16])
        print(f"N={N} {H}x{W} {ci}->{co} {mname:9s} 4-wave {a:7.3f} ms {fl / a / 1e9:6.1f} TF/s | PC {b:7.3f} ms {fl / b / 1e9:6.1f} TF/s x{a / b:5.3f} | "
              f"F8 {c:7.3f} ms {fl / c / 1e9:6.1f} TF/s x{a / c:5.3f}", flush=True)
