"""Experiment (GPU box; test infrastructure: runs the oracle): alpha error and attention time with the residual terms of the P.V
product switched off (SDM_ATTN_PV_FP16=1: P and V^T as plain fp16 operands, Q.K^T stays split) - full architecture, 512x512."""
import os, sys, time, torch
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
S = 512
img, tri = synthetic_inputs(1, S, S)
ref, _ = O.apply_matte(w, cfg.as_dict(), img, tri, S, False, "alpha_only", False, 0.8)
img4, tri4 = synthetic_inputs(4, 1024, 1024, seed=1234)
for pv in ("0", "1"):
    os.environ["SDM_ATTN_PV_FP16"] = pv
    eng = Engine(cfg, 0)
    eng.load_state_dict(w)
    a = eng.apply_matte(img.cuda(), tri.cuda(), S).cpu()
    d = (a - ref).abs()
    eng.apply_matte(img4.cuda(), tri4.cuda(), 1024)
    eng.profile(True); eng.apply_matte(img4.cuda(), tri4.cuda(), 1024); eng.profile(False)
    pr = eng.profile_results()
    print(f"SDM_ATTN_PV_FP16={pv}: max|d|={d.max().item():.3e} mean={d.mean().item():.3e}  attn_d64 {pr['attn_d64']['ms']:.2f} ms/step (B=4 1024^2)", flush=True)
    eng.close()
