"""One conv shape, a few launches, for rocprofv3 --pmc passes (bench helper).  argv: N H W Cin Cout flag iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16")
N, H, W, ci, co, flag, iters = [int(x) for x in sys.argv[1:8]]
print(eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, iters=iters))
