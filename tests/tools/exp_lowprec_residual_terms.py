"""Numerical experiment (test infrastructure: runs the fp32 CPU oracle, nothing of the product path): what alpha error remains if
the two residual terms of the split-fp16 product (x_lo * w  and  x * w_lo) are evaluated on LOWER-precision operands (OCP fp8 e4m3 /
e5m2 with fixed scales, MX block-scaled fp8 / fp6 / fp4 as gfx950's `v_mfma_scale_f32_32x32x64_f8f6f4` consumes them) while
x_hi * w_hi stays an fp16 product.  Both operands of a residual term are quantised (the f8f6f4 MFMA takes no fp16 operand).
DESIGN.md section 5 ("next lever") quotes the output: profiles/r02_lowprec_residual_terms.txt.

usage: python tests/tools/exp_lowprec_residual_terms.py [S=256]"""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.weights import synthetic_state_dict
from comfyui_sdmatte_amd.synth import synthetic_inputs
from oracle import sdmatte_oracle as O
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = SDMatteConfig.full(); w = synthetic_state_dict(cfg, 0)
img, tri = synthetic_inputs(1, S, S)
real_conv = F.conv2d; real_lin = F.linear
FMT = {"e2m3": (2, 3, 1), "e3m2": (3, 2, 3), "e2m1": (2, 1, 1), "e4m3": (4, 3, 7), "e5m2": (5, 2, 15)}
def qelem(x, fmt):
    eb, mb, bias = FMT[fmt]
    emax = (2 ** eb - 1) - bias if fmt not in ("e5m2",) else (2 ** eb - 2) - bias
    if fmt == "e4m3": vmax = 448.0
    elif fmt == "e5m2": vmax = 57344.0
    else: vmax = (2 - 2.0 ** -mb) * 2.0 ** emax
    emin = 1 - bias
    ax = x.abs().clamp_min(1e-45)
    e = torch.floor(torch.log2(ax)).clamp(emin, emax)
    step = torch.exp2(e - mb)
    return (torch.round(x / step) * step).clamp(-vmax, vmax)
def qblock(x, fmt, dim, fixed=None):
    """MX quantisation: blocks of 32 along dim share a power-of-two scale (or a fixed global scale)"""
    eb, mb, bias = FMT[fmt]
    emax = (2 ** eb - 1) - bias if fmt != "e5m2" else (2 ** eb - 2) - bias
    if fmt == "e4m3": emax = 8
    if fixed is not None:
        return qelem(x * fixed, fmt) / fixed
    xm = x.movedim(dim, -1)
    K = xm.shape[-1]; pad = (-K) % 32
    xp = F.pad(xm, (0, pad)).reshape(*xm.shape[:-1], -1, 32)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    s = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    q = qelem(xp / s, fmt) * s
    return q.reshape(*xm.shape[:-1], -1)[..., :K].movedim(-1, dim)
def split(x):
    hi = x.half().float(); return hi, x - hi
MODE = {"m": None}
def terms(x, wt, m, xdim, wdim):
    xh, xl = split(x); wh, wl = split(wt)
    k = m["kind"]
    if k == "fp16": return [(xh, wh)]
    if k == "x3": return [(xh, wh), (xl.half().float(), wh), (xh, wl.half().float())]
    f = m["fmt"]
    if m.get("fixed"):
        sx, sw = m["fixed"]
        return [(xh, wh), (qblock(xl, f, xdim, 2.0 ** (sx + 11)), qblock(wt, f, wdim, 2.0 ** sw)),
                (qblock(x, f, xdim, 2.0 ** sx), qblock(wl, f, wdim, 2.0 ** (sw + 11)))]
    return [(xh, wh), (qblock(xl, f, xdim), qblock(wt, f, wdim)), (qblock(x, f, xdim), qblock(wl, f, wdim))]
def conv_patched(x, wt, b=None, stride=1, padding=0, *a, **k):
    m = MODE["m"]
    if m is None or (wt.shape[-1] != 3 and not m.get("all")): return real_conv(x, wt, b, stride, padding, *a, **k)
    out = None
    for (xa, wa) in terms(x, wt, m, 1, 1):
        y = real_conv(xa, wa, None, stride, padding); out = y if out is None else out + y
    return out if b is None else out + b.view(1, -1, 1, 1)
def lin_patched(x, wt, b=None):
    m = MODE["m"]
    if m is None or not m.get("all"): return real_lin(x, wt, b)
    out = None
    for (xa, wa) in terms(x, wt, m, -1, 1):
        y = real_lin(xa, wa); out = y if out is None else out + y
    return out if b is None else out + b
O.F.conv2d = conv_patched; O.F.linear = lin_patched
def run(m):
    MODE["m"] = m; t0 = time.time()
    a, _ = O.apply_matte(w, cfg.as_dict(), img, tri, S, False, "alpha_only", False, 0.8)
    return a, time.time() - t0
ref, t = run(None); print("fp32 oracle", t, flush=True)
for name, m in [("fp16 3x3 only", dict(kind="fp16")), ("x3 3x3 only", dict(kind="x3")),
                ("e4m3 both operands fixed scale 3x3", dict(kind="q", fmt="e4m3", fixed=(2, 10))),
                ("e4m3 both, MX block 3x3", dict(kind="q", fmt="e4m3")),
                ("fp6 e2m3 MX 3x3", dict(kind="q", fmt="e2m3")),
                ("bf6 e3m2 MX 3x3", dict(kind="q", fmt="e3m2")),
                ("fp4 e2m1 MX 3x3", dict(kind="q", fmt="e2m1")),
                ("fp6 e2m3 MX all conv+linear", dict(kind="q", fmt="e2m3", all=True)),
                ("e4m3 MX all conv+linear", dict(kind="q", fmt="e4m3", all=True))]:
    a, t = run(m); d = (a - ref).abs()
    print(f"{name:38s} max {d.max().item():.3e} mean {d.mean().item():.3e}  ({t:.0f}s)", flush=True)
