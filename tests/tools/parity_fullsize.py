"""One-off parity record on the full SD-2.1/SDMatte architecture (synthetic weights) at the BASELINE configs:
#2 1024x1024 B=1 alpha_only (engine forward vs fp32 CPU oracle), #4 768x768 through the node post-processing
(mask_refine, trimap_constraint 0.8).  GPU box only (bench helper)."""
import sys, os, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
from comfyui_sdmatte_amd.sdmatte_nodes import refine_and_compose
from oracle import sdmatte_oracle as O
cfg = SDMatteConfig.full()
w = synthetic_state_dict(cfg, 0)
eng = Engine(cfg, 0)                     # default precision (fp16x3)
eng.load_state_dict(w)
fast = Engine(cfg, 0, precision="fp16")
fast.load_state_dict(w)
res = {}
todo = [(1024, "config2_1024"), (768, "config4_768")]
if len(sys.argv) > 1:
    todo = [t for t in todo if t[1] in sys.argv[1:]]
for S, tag in todo:
    img, tri = synthetic_inputs(1, S, S)
    t0 = time.time()
    ra, rm = O.apply_matte(w, cfg.as_dict(), img, tri, S, False, "matted_rgba", tag == "config4_768", 0.8)
    tc = time.time() - t0
    a = eng.apply_matte(img.cuda(), tri.cuda(), S).cpu()
    ms = eng.last_forward_ms()
    af = fast.apply_matte(img.cuda(), tri.cuda(), S).cpu()
    if tag == "config4_768":
        a, m = refine_and_compose(a, img, tri, "matted_rgba", True, 0.8)
        frac_flip = ((a - ra).abs() > 1e-2).float().mean().item()
    else:
        frac_flip = 0.0
    d = (a - ra).abs()
    res[tag] = {"precision": "fp16x3", "max_abs": float(d.max()), "mean_abs": float(d.mean()),
                "p999": float(d.flatten().kthvalue(int(d.numel() * 0.999)).values),
                "frac_gt_1e-2": frac_flip, "oracle_cpu_s": round(tc, 1), "gpu_ms": round(ms, 2), "gpu_ms_fp16": round(fast.last_forward_ms(), 2)}
    if tag != "config4_768":
        df = (af - ra).abs()
        res[tag]["fp16_max_abs"] = float(df.max()); res[tag]["fp16_mean_abs"] = float(df.mean())
    print(tag, res[tag], flush=True)
print(json.dumps(res))
