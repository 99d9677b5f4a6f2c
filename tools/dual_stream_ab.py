"""A/B: one engine with B = 4 images per call against TWO engines on the same GPU (own HIP stream and arena each, same weights), each with half of the
batch, driven from two host threads - the second stream's kernels fill the tails and launch gaps of the first.  Bench helper."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd import engine as E
from comfyui_sdmatte_amd.config import SDMatteConfig
from comfyui_sdmatte_amd.synth import synthetic_inputs
from comfyui_sdmatte_amd.weights import synthetic_state_dict

cfg = SDMatteConfig.full()
S, B = 1024, 4
dev = torch.device("cuda", 0)
sd = synthetic_state_dict(cfg, 0)
engs = [E.Engine(cfg, 0, precision=E.DEFAULT_PRECISION) for _ in range(2)]
for e in engs:
    e.load_state_dict(sd)
img, tri = synthetic_inputs(B, S, S, seed=1234)
img_d, tri_d = img.to(dev), tri.to(dev)
alpha = torch.empty(B, S, S, dtype=torch.float32, device=dev)


def run_single(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        engs[0].apply_matte(img_d, tri_d, S, False, out=alpha, sync=False)
    engs[0].synchronize() if hasattr(engs[0], "synchronize") else None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run_dual(n, split=(2, 2)):
    lo = [0, split[0]]
    hi = [split[0], B]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def work(k):
        with torch.cuda.stream(streams[k]):
            for _ in range(n):
                engs[k].apply_matte(img_d[lo[k]:hi[k]], tri_d[lo[k]:hi[k]], S, False, out=alpha[lo[k]:hi[k]], sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


run_single(2)
run_dual(2)
for rep in range(2):
    a = run_single(5)
    b = run_dual(5)
    c = run_dual(5, split=(1, 3))
    print(f"single engine B=4: {a * 1e3:8.2f} ms/step {B / a:7.3f} images/s | two engines B=2+2: {b * 1e3:8.2f} ms/step {B / b:7.3f} images/s | B=1+3: {c * 1e3:8.2f} ms/step", flush=True)
