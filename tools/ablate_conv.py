"""Runtime ablation of the conv kernel on the GPU box (bench helper, not part of the product path).
in_f32 flag bits: 1 fp32 activations, 2 split-precision kernel, 4 fused GroupNorm+SiLU staging."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
shapes = [("128->128 @1024^2 N=4", (4, 1024, 1024, 128, 128, 9)), ("512->512 @256^2 N=4", (4, 256, 256, 512, 512, 9))]
names = {0: "full", 1: "no A staging", 2: "no weight DMA", 3: "no A + no DMA", 4: "no MFMA sweeps", 8: "no epilogue stores", 16: "no stage barriers", 19: "MFMA only", 7: "nothing but barriers", 27: "MFMA only, no epilogue stores", 32: "no A loads (writes kept)", 64: "no A writes (loads kept)"}
modes = [("fp16 in", 0), ("fp32 in", 1), ("fp32 in + GN", 5), ("split", 3), ("split + GN", 7)]
if len(sys.argv) > 1:
    modes = [m for m in modes if sys.argv[1] in m[0]]
for label, (N, H, W, ci, co, nt) in shapes:
    fl = 2.0 * N * H * W * ci * co * nt
    for mname, flag in modes:
        if nt == 1 and (flag & 4):
            continue
        print(label, "|", mname)
        for ab in (0, 1, 32, 64, 2, 3, 16, 19, 27, 4, 8, 7):
            ms = eng.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, ablate=ab, iters=10)
            print(f"   ablate={ab:2d} {names[ab]:20s} {ms:8.4f} ms  ({fl / ms / 1e9:8.1f} TF/s algorithmic)")
