"""Numerical experiment (test infrastructure: runs the fp32 CPU oracle, nothing of the product path): what alpha error remains if the
stride-1 3x3 convolutions with >= 128 input and output channels are evaluated as Winograd F(2x2, 3x3) - 16 transform-domain GEMMs,
2.25x fewer multiplies - in the engine's arithmetic: input transform B^T d B in fp32, THEN the operand split (fp16 high part + e5m2
residual pair), weights transformed G g G^T in fp64 and split at pack time (fp16 high part + e4m3 residual pair with a per-layer
power-of-two scale), fp32 accumulation, output transform A^T m A in fp32.  Compared with the same operand formats on the direct
convolution (what the F8 kernel computes today).  DESIGN.md quotes the output: profiles/r04_winograd_numerics.txt.

usage: python tests/tools/exp_winograd.py [S=256]"""
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
real_conv = F.conv2d


def q8(x, mb, emin, emax, vmax):
    ax = x.abs().clamp_min(1e-45)
    e = torch.floor(torch.log2(ax)).clamp(emin, emax)
    step = torch.exp2(e - mb)
    return (torch.round(x / step) * step).clamp(-vmax, vmax)
e5m2 = lambda x: q8(x, 2, -14, 15, 57344.0)
e4m3 = lambda x: q8(x, 3, -6, 8, 448.0)


def operands(x, wt):
    """the three products of the engine's split arithmetic: (x_hi, w_hi), (e5m2(x_lo 2^11), e4m3(w s)), (e5m2(x), e4m3(w_lo 2^11 s))"""
    x = x.clamp(-57344.0, 57344.0)
    xh = x.half().float(); xl = x - xh
    wh = wt.half().float(); wl = wt - wh
    s = torch.exp2(torch.floor(torch.log2(448.0 / wt.abs().max().clamp_min(1e-30))))
    return [(xh, wh), (e5m2(xl * 2048.0) / 2048.0, e4m3(wt * s) / s), (e5m2(x), e4m3(wl * 2048.0 * s) / (2048.0 * s))]


Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
MODE = {"m": None}
STATS = {"layers": 0}


def winograd(x, wt, split):
    Bn, C, H, W = x.shape
    Co = wt.shape[0]
    Hp, Wp = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = F.unfold(xp, 4, stride=2).view(Bn, C, 4, 4, -1)                              # [B, C, 4, 4, T]
    V = torch.einsum("ij,bcjkt,lk->bcilt", Bt, d, Bt)                                # B^T d B in fp32
    U = torch.einsum("ij,ocjk,lk->ocil", G, wt.double(), G).float()                  # G g G^T in fp64, stored fp32 -> split below
    if split:
        M = None
        for (va, ua) in operands(V, U):
            m = torch.einsum("ocil,bcilt->boilt", ua, va)
            M = m if M is None else M + m
    else:
        M = torch.einsum("ocil,bcilt->boilt", U, V)
    Y = torch.einsum("ij,bojkt,lk->boilt", At, M, At)                                # [B, Co, 2, 2, T]
    out = F.fold(Y.reshape(Bn, Co * 4, -1), (Hp, Wp), 2, stride=2)
    return out[:, :, :H, :W]


def conv_patched(x, wt, b=None, stride=1, padding=0, *a, **k):
    m = MODE["m"]
    wide = wt.dim() == 4 and wt.shape[-1] == 3 and stride in (1, (1, 1)) and wt.shape[0] >= 128 and wt.shape[1] >= 128 and padding in (1, (1, 1))
    if m is None or not wide:
        return real_conv(x, wt, b, stride, padding, *a, **k)
    STATS["layers"] += 1
    if m == "direct_split":
        out = None
        for (xa, wa) in operands(x, wt):
            y = real_conv(xa, wa, None, 1, 1); out = y if out is None else out + y
    elif m == "wino_fp32":
        out = winograd(x, wt, False)
    else:
        out = winograd(x, wt, True)
    return out if b is None else out + b.view(1, -1, 1, 1)


O.F.conv2d = conv_patched


def run(m):
    MODE["m"] = m; STATS["layers"] = 0; t0 = time.time()
    a, _ = O.apply_matte(w, cfg.as_dict(), img, tri, S, False, "alpha_only", False, 0.8)
    return a, time.time() - t0


ref, t = run(None)
print(f"fp32 oracle at {S}x{S}: {t:.1f} s", flush=True)
for name, m in (("direct conv, split operands (fp16 hi + fp8 residual pair: the F8 kernel's arithmetic)", "direct_split"),
                ("Winograd F(2x2,3x3), fp32 operands (the transform's own rounding)", "wino_fp32"),
                ("Winograd F(2x2,3x3), split operands in the transform domain", "wino_split")):
    a, t = run(m)
    d = (a - ref).abs()
    print(f"{name}: {STATS['layers']} layers, max|d alpha| = {d.max():.3e} mean = {d.mean():.3e}  ({t:.0f} s)", flush=True)
