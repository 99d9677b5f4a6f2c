"""Runtime ablation of the d=64 attention kernel on the GPU box (bench helper, not part of the product path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
names = {0: "full", 32: "late LDS stage (old order)", 1: "no softmax", 2: "no PV", 4: "no QK", 8: "no prefetch", 3: "QK only", 6: "softmax only", 7: "staging+barrier only", 15: "LDS staging+barrier only"}
for (B, h, Lq, Lk) in [(4, 5, 16384, 16384), (4, 10, 4096, 16384)]:
    fl = 4.0 * B * h * Lq * Lk * 64
    for qt in (1, 5, 3, 7):
        print(f"B={B} h={h} Lq={Lq} Lk={Lk} QT={qt}")
        for ab in (0, 0, 1, 8, 7):
            ms = eng.bench_attn(B, h, Lq, Lk, qt=qt, ablate=ab, iters=5)
            print(f"   ablate={ab:2d} {names[ab]:26s} {ms:8.4f} ms  ({fl / ms / 1e9:8.1f} TF/s equiv)")
