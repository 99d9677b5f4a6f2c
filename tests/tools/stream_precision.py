"""fp32 vs fp16 residual stream: parity vs the fp32 oracle on the full architecture at 512^2 (GPU box only; bench helper)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
from oracle import sdmatte_oracle as O
cfg = SDMatteConfig.full()
w = synthetic_state_dict(cfg, 0)
img, tri = synthetic_inputs(1, 512, 512)
data = O.preprocess(img, tri, 512, False)
ref = O.sdmatte_forward(w, cfg.as_dict(), data)
for sf in (True, False):
    eng = Engine(cfg, 0, stream_f32=sf)
    eng.load_state_dict(w)
    out = eng.forward(data["image"].cuda(), data["trimap"].cuda()).cpu()
    d = (out - ref).abs()
    print(f"stream_f32={sf}: max|d|={d.max():.3e} mean|d|={d.mean():.3e}")
    eng.close()
