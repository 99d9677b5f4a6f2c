"""Compile-time ablations of the d=512 single-head attention kernel on the GPU box (sdm_bench_attn, qt bit 64).  Bench helper."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16x3")
names = {0: "full", 1: "no exchange / softmax", 6: "no MFMAs", 7: "no MFMAs, no softmax (DMAs + fragment reads + barriers)", 8: "no DMAs", 32: "no fragment reads",
         40: "no DMAs, no fragment reads", 41: "MFMAs + barriers only"}
for (B, Lq, Lk) in [(8, 16384, 16384), (4, 16384, 16384)]:
    fl = 4.0 * B * Lq * Lk * 512
    print(f"B={B} Lq={Lq} Lk={Lk} d=512")
    for ab in (0, 0, 1, 6, 7, 8, 32, 40, 41):
        ms = eng.bench_attn(B, 1, Lq, Lk, qt=64, ablate=ab, iters=3)
        print(f"   ablate={ab:2d} {names[ab]:60s} {ms:8.4f} ms  ({fl / ms / 1e9:8.1f} TF/s equiv)")
eng.close()
