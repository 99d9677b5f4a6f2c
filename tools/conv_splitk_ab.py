"""A/B of split-K (engine option conv_splitk = 0 / 2 / 4 / 8) for the register-staged conv / GEMM kernels on the low-resolution U-Net shapes, in the
engine's real I/O format (fp32 in / out, split-precision operands, residual, statistics; random operands), at 1 and 4 images per call.
The timed region holds the partial-sum launch AND splitk_reduce_kernel.  Bench helper."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
F = 1 | 2 | 32 | 64 | 128          # fp32 in, split precision, fp32 out, fp32 residual, statistics
# (ntaps, H, W, Cin, Cout, tile cfg or -1 = the engine's choice)
shapes = [(9, 16, 16, 1280, 1280, -1), (9, 16, 16, 2560, 1280, -1), (9, 32, 32, 1280, 1280, 1), (9, 32, 32, 2560, 1280, 1), (9, 64, 64, 640, 640, 1),
          (9, 64, 64, 1280, 640, 1),
          (1, 16, 16, 1280, 1280, -1), (1, 16, 16, 5120, 1280, -1), (1, 16, 16, 2560, 1280, -1), (1, 32, 32, 1280, 1280, -1), (1, 32, 32, 5120, 1280, -1),
          (1, 32, 32, 2560, 1280, -1), (1, 64, 64, 640, 640, -1), (1, 64, 64, 2560, 640, -1), (1, 128, 128, 320, 320, -1), (1, 128, 128, 1280, 320, -1)]
for N in (1, 4):
    for (nt, H, W, ci, co, cfg) in shapes:
        fl = 2.0 * N * H * W * ci * co * nt
        res = {}
        for rep in range(3):
            for ks in (0, 2, 4, 8):
                eng.lib.set_option("conv_splitk", ks)
                # GEMMs with K >= 1024 take the fp8-residual kernel in the engine (not split): bit 4 off here = the register-staged kernel
                t = eng.bench_conv(N, H, W, ci, co, ntaps=nt, in_f32=F, tile_cfg=cfg, iters=20)
                res[ks] = min(res.get(ks, 1e9), t)
        best = min(res, key=lambda k: res[k])
        print(f"N={N} {nt}tap {H}x{W} {ci}->{co} cfg {cfg:2d}: " + " | ".join(f"ks{k} {res[k] * 1000:7.1f} us {fl / res[k] / 1e9:6.1f} TF" for k in (0, 2, 4, 8)) +
              f" | best ks{best} x{res[0] / res[best]:.2f}", flush=True)
eng.lib.set_option("conv_splitk", -1)
