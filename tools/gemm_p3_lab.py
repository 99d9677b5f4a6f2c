"""Lab: what bounds the plane-fed GEMM (k_gemm.h)?  Row tile x LDS stages x ablation (1 no MFMAs, 2 no DMAs behind the prologue, 4 no epilogue) at a few
transformer shapes of a 1024^2 batch of 4.  ms per launch; results of ablated runs are garbage by design."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_sdmatte_amd.config import SDMatteConfig  # noqa: E402
from comfyui_sdmatte_amd.engine import Engine  # noqa: E402

eng = Engine(SDMatteConfig.tiny(), 0, True)
shapes = [("L0 square", 65536, 320, 320, 0), ("L0 ff1", 65536, 320, 2560, 1), ("L0 ff2-like", 65536, 1280, 320, 0), ("L1 ff1", 16384, 640, 5120, 1),
          ("L2 square", 4096, 1280, 1280, 0), ("L2 ff2-like", 4096, 5120, 1280, 0), ("L2 ff1", 4096, 1280, 10240, 1)]
cfgs = [(256, 2), (256, 3), (256, 4), (128, 2), (128, 4), (128, 5), (64, 2), (64, 4), (64, 7)]
abl = [0, 1, 2, 4, 3, 5, 6]
print("ms per launch; columns = ablate flags " + " ".join(f"a{a}" for a in abl))
for name, M, K, N, epi in shapes:
    print(f"-- {name}: M={M} K={K} N={N} epi={epi}   (2 MFMA-units/product floor at 2.5 PF: {2 * 2e-9 * M * K * N / 2.5e6:.4f} ms)")
    for bm, ns in cfgs:
        eng.lib.set_option("gemm_p3_tile", bm)
        eng.lib.set_option("gemm_p3_stages", ns)
        row = []
        for a in abl:
            eng.lib.set_option("gemm_p3_ablate", a)
            row.append(eng.bench_gemm_p3(M, K, N, epi, False, iters=20))
        print(f"   tile {bm:3d} stages {ns}: " + " ".join(f"{r:8.4f}" for r in row))
eng.lib.reset_options()
eng.close()
