"""Timeline of the F8 3x3 conv kernel's first blocks (bench helper): shader-clock stamps of consumer wave 0 / producer wave 0 per tile -
tile start, prologue done, past the first barrier, end of chunks 0-3 and of the last chunk, tile-end barrier, epilogue issued -
relative to the first stamp of block 0.  usage: python tools/conv_trace.py [N H W Cin Cout res]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
import ctypes
from comfyui_sdmatte_amd import build as B
from comfyui_sdmatte_amd.engine import Bindings, Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
lib = B.build_all(extra_flags=("-DSDM_CONV_TRACE",), out=os.path.join(B.CSRC, "libsdmatte_hip_trace.so"))      # prebuilt in the build container
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(lib)))
eng._on_device = True
a = [int(x) for x in sys.argv[1:7]] if len(sys.argv) >= 7 else [8, 1024, 1024, 128, 128, 1]
N, H, W, ci, co, res = a
flag = 1 | 2 | 16 | 32 | 128 | 4 | (64 if res else 0)
for x in ("1", "0"):
    os.environ["SDM_CONV_XTILE"] = x
    sys.stderr.write(f"== SDM_CONV_XTILE={x}  N={N} {H}x{W} {ci}->{co} res={res}\n"); sys.stderr.flush()
    ms = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=0, ablate=256, iters=4)
    sys.stderr.write(f"   {ms:.3f} ms per launch\n"); sys.stderr.flush()
