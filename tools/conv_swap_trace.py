"""Step timeline of the role-swapping F8 3x3 kernel (bench helper, -DSDM_CONV_TRACE build): wave 0 of whichever group is PRODUCING parks
(shader clock, code) at the arrival at / release from every step barrier; code 0 = a plain producer step, 1 / 2 = a step of chunk 0 / 1 of a
producer that also drains the previous tile's accumulators, 3 = released.  Prints the mean work (release -> arrival) and wait (arrival ->
release) per step kind and position.  usage: python tools/conv_swap_trace.py [N H W Cin Cout res]"""
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    import ctypes
    from comfyui_sdmatte_amd import build as B
    from comfyui_sdmatte_amd.engine import Bindings, Engine
    from comfyui_sdmatte_amd.config import SDMatteConfig
    lib = B.build_all(extra_flags=("-DSDM_CONV_TRACE",), out=os.path.join(B.CSRC, "libsdmatte_hip_trace.so"))      # prebuilt in the build container
    eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(lib)))
    eng._on_device = True
    N, H, W, ci, co, res = a
    flag = 1 | 2 | 16 | 32 | 128 | 4 | (64 if res else 0)
    ms = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=0, ablate=256, iters=4)
    sys.stderr.write(f"[ms] {ms:.4f}\n")


def analyse(text, nch):
    ms = float(re.search(r"\[ms\] ([\d.]+)", text).group(1))
    work = {}
    wait = {}
    tiles = 0
    for m in re.finditer(r"\[trace\] block (\d+) (consumer|producer) n=\d+: ([\d ]+)", text):
        v = [int(x) for x in m.group(3).split()]
        ev = [(x >> 2, x & 3) for x in v]
        # pairs (arrive code c, release 3); position inside a tile = running index of arrivals since the last code change to 1 or since start
        pos = 0
        prev_rel = None
        last_code = None
        for i in range(0, len(ev) - 1, 2):
            (ta, ca), (tr, cr) = ev[i], ev[i + 1]
            if cr != 3:
                break
            if ca == 1 and last_code != 1:
                pos = 0
            if ca == 0 and last_code not in (0, 2) and last_code is not None and last_code != 1:
                pos = 0
            key = (ca, pos % 6 if ca != 0 else pos % 6)
            if prev_rel is not None and (ta - prev_rel) % (1 << 30) < 200000:
                work.setdefault(key, []).append((ta - prev_rel) % (1 << 30))
            wait.setdefault(key, []).append((tr - ta) % (1 << 30))
            prev_rel = tr
            last_code = ca
            pos += 1
    mean = lambda x: sum(x) / len(x) if x else float("nan")
    print(f"  {ms:.3f} ms per launch")
    for code, name in ((1, "draining, chunk 0"), (2, "draining, chunk 1"), (0, "plain producer  ")):
        print(f"  {name}: work per step k=0..5 " + " ".join(f"{mean(work.get((code, k), [])):6.0f}" for k in range(6)) +
              "   wait " + " ".join(f"{mean(wait.get((code, k), [])):6.0f}" for k in range(6)) + f"   (n={len(wait.get((code, 0), []))})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(x) for x in sys.argv[2:8]])
        sys.exit(0)
    cases = [[int(x) for x in sys.argv[1:7]]] if len(sys.argv) >= 7 else [[4, 1024, 1024, 128, 128, 0], [4, 1024, 1024, 128, 128, 1], [4, 256, 256, 512, 512, 1]]
    for a in cases:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [str(x) for x in a], capture_output=True, text=True)
        print(f"== N={a[0]} {a[1]}x{a[2]} {a[3]}->{a[4]} res={a[5]}", flush=True)
        try:
            analyse(r.stderr, a[3] // 32)
        except Exception as e:
            print("  analysis failed:", e, r.stderr[-600:])
        if os.environ.get("SDM_TRACE_RAW"):
            print(r.stderr[:6000])
