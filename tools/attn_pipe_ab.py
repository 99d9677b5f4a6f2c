"""A/B of the two-tile software pipeline of the 8-wave d=64 attention kernel (default; option attn_pipe = 0 = the plain 8-wave kernel):
bit-identity of the results on level-0 / level-1 shapes (dense and with a trimap-like key bias, repeated to expose races), then the whole
step.  Bench helper.  usage: python tools/attn_pipe_ab.py"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
from comfyui_sdmatte_amd.engine import Engine
from comfyui_sdmatte_amd.config import SDMatteConfig
eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16x3")
g = torch.Generator(device="cuda").manual_seed(3)
eng.lib.set_option("attn_nw", 8)
ok = True
for (B, h, Lq, Lk) in ((2, 5, 16384, 16384), (1, 10, 1000, 4096 + 37)):
    q = torch.randn(B, Lq, h * 64, generator=g, device="cuda") * 1.5
    k = torch.randn(B, Lk, h * 64, generator=g, device="cuda") * 1.5
    v = torch.randn(B, Lk, h * 64, generator=g, device="cuda")
    bias = torch.where(torch.rand(B, Lk, generator=g, device="cuda") < 0.4, torch.tensor(-10000.0, device="cuda"), torch.tensor(0.0, device="cuda"))
    blocks = bias.clone(); blocks[:, : Lk // 3] = -10000.0            # whole key tiles masked: the active-tile list is exercised
    for name, bb in (("dense", None), ("bias", bias), ("masked tiles", blocks)):
        eng.lib.set_option("attn_pipe", 0)
        ref = eng.op_attention_split(q, k, v, h, bias=bb)
        eng.lib.set_option("attn_pipe", 1)
        outs = [eng.op_attention_split(q, k, v, h, bias=bb) for _ in range(3)]
        eng.lib.set_option("attn_pipe", 0)
        same = all(torch.equal(o, ref) for o in outs)
        ok &= same
        print(f"B={B} h={h} Lq={Lq} Lk={Lk} {name:12s} pipe == shipped (3 runs): {same}" + ("" if same else f"  max|d|={max((o - ref).abs().max().item() for o in outs):.3e}"), flush=True)
eng.lib.set_option("attn_nw", 0)
print("8-wave pipeline bit-identical:", ok)
# the 4-wave pipeline (option attn_pipe4, off by default)
eng.lib.set_option("attn_nw", 4)
ok4 = True
for (B, h, Lq, Lk) in ((1, 10, 4096, 16384), (1, 10, 1000, 4096 + 37)):
    q = torch.randn(B, Lq, h * 64, generator=g, device="cuda") * 1.5
    k = torch.randn(B, Lk, h * 64, generator=g, device="cuda") * 1.5
    v = torch.randn(B, Lk, h * 64, generator=g, device="cuda")
    blocks = torch.where(torch.rand(B, Lk, generator=g, device="cuda") < 0.4, torch.tensor(-10000.0, device="cuda"), torch.tensor(0.0, device="cuda"))
    blocks[:, : Lk // 3] = -10000.0
    for name, bb in (("dense", None), ("masked tiles", blocks)):
        eng.lib.set_option("attn_pipe4", 0)
        ref = eng.op_attention_split(q, k, v, h, bias=bb)
        eng.lib.set_option("attn_pipe4", 1)
        outs = [eng.op_attention_split(q, k, v, h, bias=bb) for _ in range(3)]
        eng.lib.set_option("attn_pipe4", 0)
        same = all(torch.equal(o, ref) for o in outs)
        ok4 &= same
        print(f"4-wave: B={B} h={h} Lq={Lq} Lk={Lk} {name:12s} pipe4 == shipped (3 runs): {same}", flush=True)
eng.lib.set_option("attn_nw", 0)
print("4-wave pipeline bit-identical:", ok4)
for name, opts in (("plain 8-wave kernel (attn_pipe=0)", ["attn_pipe=0"]), ("8-wave pipeline only (attn_pipe4=0)", ["attn_pipe4=0"]), ("default", [])):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--timed-only", "--steps", "4", "--warmup", "2"] + [x for o in opts for x in ("--opt", o)], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{name}: {d['value']} img/s {d['ms_per_step']} ms/step", {k: x["ms"] for k, x in list(d["kernel_breakdown_ms"].items())[:4]}, flush=True)
    except Exception as e:
        print("bench failed", e, r.stderr[-400:])
