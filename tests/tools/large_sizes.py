"""One-off check (GPU box): inference sizes beyond the node's 1024 on the full architecture - 1536x1536 and 1024x1536 run
directly (flash attention + 288 GB of HBM, no tiling), 2048x2048 must fail loudly (a per-image operand exceeds the 4 GB a buffer
descriptor addresses).  Test helper, not part of the product path."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
cfg = SDMatteConfig.full()
eng = Engine(cfg, 0)
eng.load_state_dict(synthetic_state_dict(cfg, 0))
for (H, W) in ((1536, 1536), (1024, 1536)):
    img, tri = synthetic_inputs(1, H, W, seed=3)
    x = ((img.permute(0, 3, 1, 2).contiguous() - 0.5) / 0.5).cuda(); t = (tri.unsqueeze(1) * 2 - 1).cuda()
    a = eng.forward(x, t).cpu(); ms = eng.last_forward_ms()
    b = eng.forward(x, t).cpu()
    print(H, W, "ms", round(ms, 1), "finite", bool(torch.isfinite(a).all()), "range", float(a.min()), float(a.max()), "std", float(a.std()), "deterministic", bool(torch.equal(a, b)))
try:
    img, tri = synthetic_inputs(1, 2048, 2048, seed=3)
    x = ((img.permute(0, 3, 1, 2).contiguous() - 0.5) / 0.5).cuda(); t = (tri.unsqueeze(1) * 2 - 1).cuda()
    eng.forward(x, t)
    print("2048 ran")
except RuntimeError as e:
    print("2048:", str(e)[:200])
