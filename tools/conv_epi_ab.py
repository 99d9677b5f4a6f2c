"""A/B of the F8 conv / GEMM kernel's epilogue on the layer shapes that dominate the step, in the engine's real I/O format
(fp32 activations in, fp32 out, optional fp32 residual, GroupNorm statistics of the consumer; random operands).
option conv_epi: 0 = LDS-staged epilogue, 3 = 0 + residual as accumulator init, 4 = 3 + register-direct 16-byte stores (default).
(profiles/r03_conv_epilogue_ab.txt holds the round-3 measurement of the abandoned dword-store variants 1 / 2.)
Bench helper, not part of the product path.  usage: python tools/conv_epi_ab.py [quick]
flag bits: 1 fp32 in, 2 split, 4 fused GroupNorm+SiLU, 16 fp8 residual terms, 32 fp32 out, 64 fp32 residual, 128 statistics"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
quick = len(sys.argv) > 1 and sys.argv[1].startswith("quick")
MODES = ("0", "3", "4")
shapes = [(8, 1024, 1024, 128, 128, 9), (4, 1024, 1024, 128, 128, 9), (8, 512, 512, 256, 256, 9), (8, 256, 256, 512, 512, 9), (4, 1024, 1024, 256, 128, 9),
          (4, 128, 128, 320, 320, 9), (4, 64, 64, 640, 640, 9), (4, 32, 32, 1280, 1280, 9),
          (4, 128, 128, 1280, 320, 1), (4, 64, 64, 2560, 640, 1), (4, 32, 32, 5120, 1280, 1)]
if quick:
    shapes = shapes[:4]
BASE = 1 | 2 | 16 | 32 | 128                      # fp32 in, split, F8, fp32 out, statistics
for (N, H, W, ci, co, nt) in shapes:
    fl = 2.0 * N * H * W * ci * co * nt
    for name, flag in (("conv1 (GN, no res)", BASE | 4), ("conv2 (GN, res)", BASE | 4 | 64)) if nt == 9 else (("gemm (res)", BASE | 64),):
        res = {}
        for rep in range(2):
            for mode in MODES:
                eng.lib.set_option("conv_epi", int(mode))
                t = eng.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, tile_cfg=0 if nt == 9 else 4, iters=6)
                res[mode] = min(res.get(mode, 1e9), t)
        eng.lib.set_option("conv_epi", int(MODES[-1]))
        tn = eng.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=flag, tile_cfg=0 if nt == 9 else 4, ablate=8, iters=6)      # no stores at all
        s = " | ".join(f"epi{m} {res[m]:7.3f} ms {fl / res[m] / 1e9:6.1f} TF/s" for m in MODES)
        print(f"N={N} {H}x{W} {ci}->{co} taps={nt} {name:19s} {s} | x{res['0'] / res[MODES[-1]]:5.3f} | no-store {tn:7.3f} ms", flush=True)
if not quick:
    print("== tiles per block (option conv_f8_tpb) with option conv_epi=4, conv2 form")
    eng.lib.set_option("conv_epi", 4)
    for (N, H, W, ci, co, nt) in shapes[:5]:
        fl = 2.0 * N * H * W * ci * co * nt
        out = []
        for tpb in ("1", "2", "4"):
            eng.lib.set_option("conv_f8_tpb", int(tpb))
            t = min(eng.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=BASE | 4 | 64, tile_cfg=0, iters=6) for _ in range(2))
            out.append(f"tpb{tpb} {t:7.3f} ms {fl / t / 1e9:6.1f}")
        eng.lib.set_option("conv_f8_tpb", 0)
        print(f"N={N} {H}x{W} {ci}->{co}: " + " | ".join(out), flush=True)
