import sys, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd import engine as E
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
cfg = SDMatteConfig.full()
eng = E.Engine(cfg, 0, precision=E.DEFAULT_PRECISION)
eng.load_state_dict(synthetic_state_dict(cfg, 0))
img, tri = synthetic_inputs(2, 1024, 1024, 1234)
img[1] = img[0]; tri[1] = tri[0]
res = {}
for name, opts in (("base", {}), ("skip0", {"trimap_skip": 0}), ("skip0_tpb1", {"trimap_skip": 0, "conv_f8_tpb": 1}), ("skip0_xtile0", {"trimap_skip": 0, "conv_xtile": 0}),
                   ("skip0_splitk0", {"trimap_skip": 0, "conv_splitk": 0}), ("skip0_dense", {"trimap_skip": 0, "attn_dense": 1})):
    eng.lib.reset_options()
    for k, v in opts.items(): eng.lib.set_option(k, v)
    a1 = eng.apply_matte(img[:1].cuda(), tri[:1].cuda(), 1024, False).cpu()
    a2 = eng.apply_matte(img.cuda(), tri.cuda(), 1024, False).cpu()
    res[name] = a1
    print(f"{name:14s} B=1 vs B=2[0]: {(a1[0]-a2[0]).abs().max():.3e}  B=2[0] vs B=2[1]: {(a2[0]-a2[1]).abs().max():.3e}", flush=True)
for k in res:
    if k != "skip0": print(f"{k:14s} vs skip0 (B=1): {(res[k]-res['skip0']).abs().max():.3e}")
