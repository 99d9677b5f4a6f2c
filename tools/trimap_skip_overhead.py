import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd import engine as E
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.synth import synthetic_inputs
from comfyui_sdmatte_amd.weights import synthetic_state_dict
cfg = SDMatteConfig.full(); S, B = 1024, 4
dev = torch.device("cuda", 0)
eng = E.Engine(cfg, 0, precision=E.DEFAULT_PRECISION)
eng.load_state_dict(synthetic_state_dict(cfg, 0))
img, tri = synthetic_inputs(B, S, S, seed=1234)
alpha = torch.empty(B, S, S, dtype=torch.float32, device=dev)
tris = {"disc": tri, "noise": torch.rand(B, S, S), "all-background": torch.zeros(B, S, S)}
def run(t, n=5):
    td = t.to(dev); imd = img.to(dev)
    for _ in range(2): eng.apply_matte(imd, td, S, False, out=alpha, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.apply_matte(imd, td, S, False, out=alpha, sync=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, t in tris.items():
    r = {}
    for rep in range(2):
        for sk in (1, 0):
            eng.lib.set_option("trimap_skip", sk)
            r[sk] = min(r.get(sk, 1e9), run(t))
    print(f"{name:15s} trimap_skip=1 {r[1]:8.2f} ms | trimap_skip=0 {r[0]:8.2f} ms | delta {r[1]-r[0]:+.2f}", flush=True)
