"""Runtime ablation of the conv kernel on the GPU box (bench helper, not part of the product path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0)
shapes = [("128->128 @1024^2 N=8", (8, 1024, 1024, 128, 128, 9)), ("512->512 @256^2 N=8", (8, 256, 256, 512, 512, 9)),
          ("gemm 1024->2560 M=32768", (2, 128, 128, 1024, 2560, 1)), ("gemm 320->320 M=32768", (2, 128, 128, 320, 320, 1))]
names = {0: "full", 1: "no global loads", 2: "no LDS writes", 3: "no loads+writes", 4: "no MFMA phase", 8: "no epilogue stores", 7: "barriers only", 15: "nothing"}
for label, (N, H, W, ci, co, nt) in shapes:
    fl = 2.0 * N * H * W * ci * co * nt
    print(label)
    for ab in (0, 1, 2, 3, 4, 8, 7, 15):
        ms = eng.bench_conv(N, H, W, ci, co, ntaps=nt, ablate=ab, iters=20)
        print(f"   ablate={ab:2d} {names[ab]:20s} {ms:8.4f} ms  ({fl / ms / 1e9:8.1f} TF/s equiv)")
