"""Compile-time ablations of the ping-pong d=64 attention kernel on the GPU box (sdm_bench_attn, qt bit 16): which part of a phase costs what.  Bench helper."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16x3")
names = {0: "full (one barrier per tile and wave)", 1: "no softmax VALU", 6: "no MFMAs", 7: "no MFMAs, no softmax (DMAs + fragment reads + barriers)", 8: "no DMAs", 24: "no DMAs (24)",
         32: "no fragment reads", 56: "no DMAs / fragment reads (MFMAs + softmax + barriers)", 63: "barriers only", 64: "full + segment stamps (see stderr lines)", 110: "DS = 0: every DMA at the end of the matrix segment", 111: "two barriers per tile (OB = 0), DS = 1", 112: "DS = 2", 113: "DS = 3: every DMA at the head of the softmax segment",
         114: "DS = 0 + stamps", 116: "DS = 2 + stamps", 117: "DS = 3 + stamps",
         100: "KE = -1: four slots, DMAs + K reads inside the matrix segment (first DMA build)", 101: "KE = 2: DMAs AND every K read in the softmax segment",
         102: "KE = 1: + first K half read in the softmax segment"}
for (B, h, Lq, Lk) in [(4, 5, 16384, 16384), (4, 10, 4096, 16384)]:
    fl = 4.0 * B * h * Lq * Lk * 64
    tiles = (Lk // 64)
    for prio in (0,):
        print(f"B={B} h={h} Lq={Lq} Lk={Lk} " + {0: "static priority for waves 4-7", 32: "equal priorities", 128: "per-segment priority flips (matrix segments at 2)", 256: "static priority + packed fp32 subtractions"}[prio])
        for ab in (0, 0, 111, 111, 0, 111, 64, 1, 6, 8, 32, 56):
            ms = eng.bench_attn(B, h, Lq, Lk, qt=22 | prio, ablate=ab, iters=5)
            # cycles per (tile, block) at a nominal 2.1 GHz: blocks per CU = B*h*Lq/256/256
            rounds = B * h * Lq / 256 / 256
            cyc = ms * 1e-3 * 2.1e9 / (rounds * tiles)
            print(f"   ablate={ab:2d} {names[ab]:70s} {ms:8.4f} ms  ({fl / ms / 1e9:8.1f} TF/s equiv, ~{cyc:6.0f} cycles per key tile and block)")
    # the two-tile pipeline it replaces, same operands
eng.close()
