"""Per-shape timing of the plane-fed GEMM (k_gemm.h) at the transformer blocks' shapes of a 1024^2 batch of 4, next to the register-staged 1x1 kernel
(k_conv.h) the same layers ran on before.  usage: python tools/gemm_p3_bench.py [--tiles]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_sdmatte_amd.config import SDMatteConfig  # noqa: E402
from comfyui_sdmatte_amd.engine import Engine  # noqa: E402

eng = Engine(SDMatteConfig.tiny(), 0, True)
B = 4
shapes = []
for lvl, (C, hw) in enumerate(((320, 128), (640, 64), (1280, 32), (1280, 16))):
    M = B * hw * hw
    shapes += [(f"L{lvl} proj_in/out,o1,o2 (x5)", M, C, C, 0, True), (f"L{lvl} qkv", M, C, 3 * C, 2, False), (f"L{lvl} q2", M, C, C, 2, False),
               (f"L{lvl} ff1 geglu", M, C, 8 * C, 1, False), (f"L{lvl} ff2", M, 4 * C, C, 3, True), (f"L{lvl} proj_out stats", M, C, C, 4, True)]
tiles = (0, 256, 128, 64) if "--tiles" in sys.argv else (0,)
print(f"{'shape':28s} {'M':>7s} {'K':>5s} {'N':>6s} epi " + " ".join(f"{'t' + str(t):>9s}" for t in tiles) + "   TF/s(best)  old_ms")
tot = {t: 0.0 for t in tiles}
tot_old = 0.0
for name, M, K, N, epi, res in shapes:
    row = []
    for t in tiles:
        eng.lib.set_option("gemm_p3_tile", t)
        ms = eng.bench_gemm_p3(M, K, N, epi, res, iters=20)
        row.append(ms)
        tot[t] += ms * (5 if "(x5)" in name else 1)
    eng.lib.set_option("gemm_p3_tile", 0)
    hw = int((M // B) ** 0.5)
    old = eng.bench_conv(B, hw, hw, K, N, ntaps=1, in_f32=1 | 2 | 32 | (64 if res else 0), iters=20)
    tot_old += old * (5 if "(x5)" in name else 1)
    best = min(r for r in row if r > 0)
    print(f"{name:28s} {M:7d} {K:5d} {N:6d} {epi:3d} " + " ".join(f"{r:9.4f}" for r in row) + f"   {2e-9 * M * K * N / best:9.1f}  {old:7.4f}")
print("sum over one transformer block per level (ms): " + " ".join(f"t{t}={v:.3f}" for t, v in tot.items()) + f"  old={tot_old:.3f}")
eng.close()
