"""Experiment: error of the precise engine vs the fp32 oracle and vs an fp64 evaluation of the same oracle, tiny architecture, growing S."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd import engine as E
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
from oracle import sdmatte_oracle as O
cfg = SDMatteConfig.tiny()
w = synthetic_state_dict(cfg, 0)
w64 = {k: v.double() for k, v in w.items()}
for S in [int(x) for x in sys.argv[1:]] or [256, 512, 768]:
    img, tri = synthetic_inputs(1, S, S, seed=101)
    data = O.preprocess(img, tri, S, False)
    ref = O.sdmatte_forward(w, cfg.as_dict(), data)
    # fp64 evaluation of the same restatement (.float() calls inside the oracle are patched out)
    d64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in data.items()}
    import torch.nn.functional as F
    of, ol, oc = torch.Tensor.float, F.linear, F.conv2d
    torch.Tensor.float = lambda self: self.double()
    F.linear = lambda x, ww, b=None: ol(x.to(ww.dtype), ww, b)
    F.conv2d = lambda x, ww, b=None, **kw: oc(x.to(ww.dtype), ww, b, **kw)
    try:
        ref64 = O.sdmatte_forward(w64, cfg.as_dict(), d64)
    finally:
        torch.Tensor.float, F.linear, F.conv2d = of, ol, oc
    print(f"S={S}: |oracle32 - oracle64| max {float((ref.double() - ref64).abs().max()):.3e}", flush=True)
    for name, mask, env in (("fp16", 0, {}), ("fp16x3", 63, {}), ("fp16x3 no-gn-fuse", 63, {"SDM_NO_GN_FUSE": "1"})):
        for k, v in env.items():
            os.environ[k] = v
        eng = E.Engine(cfg, 0, precision=mask)
        eng.load_state_dict(w)
        out = eng.forward(data["image"].cuda(), data["trimap"].cuda(), is_trans=data["is_trans"].numpy()).cpu()
        eng.close()
        for k in env:
            del os.environ[k]
        d32 = (out - ref).abs(); d64e = (out.double() - ref64).abs()
        print(f"   {name:20s} vs oracle32: max {float(d32.max()):.3e} rms {float(d32.pow(2).mean().sqrt()):.3e} | vs oracle64: max {float(d64e.max()):.3e} rms {float(d64e.pow(2).mean().sqrt()):.3e}", flush=True)
