"""Timeline of the F8 3x3 conv kernel (bench helper, -DSDM_CONV_TRACE build): LDS-parked shader-clock stamps of consumer wave 0 / producer
wave 0 of 16 blocks (ConvParams::trace, k_conv.h) - per tile: tile start, arrival at / release from the first barrier, [producer: arrival
at / release from the barrier of every step], tile-end barrier passed, [consumer: epilogue issued].  Prints per layer class the average
producer work / wait per step kind and the consumers' out-of-loop phases.
usage: python tools/conv_trace.py [N H W Cin Cout res gn [skip [block0/256]]]"""
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
    # bench-only build, kept out of the package directory (prebuilt in the build container; SDM_TRACE_LIB names another traced build, e.g. of an older tree)
    os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
    lib = os.environ.get("SDM_TRACE_LIB") or B.build_all(extra_flags=("-DSDM_CONV_TRACE",), out=os.path.join(ROOT, "tools", "_build", "libsdmatte_hip_trace.so"))
    eng = Engine(SDMatteConfig.tiny(), 0, precision="fp16", _lib=Bindings(ctypes.CDLL(lib)))
    eng._on_device = True
    N, H, W, ci, co, res, gn, skip, b0 = a
    flag = 1 | 2 | 16 | 32 | 128 | (4 if gn else 0) | (64 if res else 0)
    ms = eng.bench_conv(N, H, W, ci, co, ntaps=9, in_f32=flag, tile_cfg=0, ablate=256 | (skip << 9) | (b0 << 12), iters=4)
    sys.stderr.write(f"[ms] {ms:.4f}\n")


def analyse(text, nch):
    tr = {}
    for m in re.finditer(r"\[trace\] block (\d+) (consumer|producer) n=\d+: ([\d ]+)", text):
        tr[(int(m.group(1)), m.group(2))] = [int(x) for x in m.group(3).split()]
    ms = float(re.search(r"\[ms\] ([\d.]+)", text).group(1))

    def tiles(v):
        out, cur = [], None
        for x in v:
            if x & 1:
                cur = []
                out.append(cur)
            if cur is not None:
                cur.append(x & ~1)
        return out
    import statistics as st
    work = [[[] for _ in range(6)] for _ in range(2)]      # [first chunk / later chunks][k]
    wait = [[[] for _ in range(6)] for _ in range(2)]
    pfirst_w, ptile = [], []
    for (b, r), v in tr.items():
        if r != "producer":
            continue
        for t in tiles(v):
            if len(t) != 4 + 12 * nch:
                continue
            pfirst_w.append(t[2] - t[1])
            prev = t[2]
            for s in range(6 * nch):
                arr, rel = t[3 + 2 * s], t[4 + 2 * s]
                i = 0 if s < 6 else 1
                work[i][s % 6].append(arr - prev)
                wait[i][s % 6].append(rel - arr)
                prev = rel
            ptile.append(t[-1] - t[0])
    cphase = [[] for _ in range(5)]
    esub = [[] for _ in range(5)]
    for (b, r), v in tr.items():
        if r != "consumer":
            continue
        ts = tiles(v)
        for i, t in enumerate(ts):
            if len(t) != 9:
                continue
            cphase[0].append(t[1] - t[0])      # tile start -> arrival at the first barrier (residual-init loads issued and landed)
            cphase[1].append(t[2] - t[1])      # wait at the first barrier
            cphase[2].append(t[3] - t[2])      # chunk loop + tile-end barrier
            cphase[3].append(t[8] - t[3])      # epilogue issue
            for q in range(4):
                esub[q].append(t[4 + q] - t[3 + q])      # sub-tile q: bias add, 8 stores, statistics accumulation
            esub[4].append(t[8] - t[7])        # statistics reduction
            if i + 1 < len(ts) and len(ts[i + 1]) == 9:
                cphase[4].append(ts[i + 1][0] - t[8])
    m = lambda x: (sum(x) / len(x)) if x else float("nan")
    print(f"  {ms:.3f} ms per launch; traced producer tiles {len(ptile)}, mean tile {m(ptile):.0f} cycles; MFMA issue per step 1536, per tile {1536 * 6 * nch}")
    for i, nm in enumerate(("chunk 0   ", "chunks 1.. ")):
        print(f"  producer {nm} work per step k=0..5: " + " ".join(f"{m(work[i][k]):6.0f}" for k in range(6)) +
              "   wait: " + " ".join(f"{m(wait[i][k]):6.0f}" for k in range(6)))
    tot_work = sum(m(work[1][k]) for k in range(6)); tot_wait = sum(m(wait[1][k]) for k in range(6))
    print(f"  producer per later chunk: work {tot_work:.0f} + wait {tot_wait:.0f} = {tot_work + tot_wait:.0f} cycles (MFMA issue 9216)")
    print(f"  producer wait at the first barrier of a tile: {m(pfirst_w):.0f}")
    print("  consumer: start->first barrier {:.0f} | wait there {:.0f} | chunk loop {:.0f} | epilogue issue {:.0f} | gap to next tile {:.0f}".format(*[m(x) for x in cphase]))
    print("  consumer epilogue: sub-tiles {:.0f} {:.0f} {:.0f} {:.0f} | statistics reduction {:.0f}".format(*[m(x) for x in esub]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(x) for x in sys.argv[2:11]])
        sys.exit(0)
    cases = [[int(x) for x in sys.argv[1:]]] if len(sys.argv) >= 8 else [
        [4, 1024, 1024, 128, 128, 0, 1], [4, 1024, 1024, 128, 128, 1, 1], [8, 512, 512, 256, 256, 1, 1], [8, 256, 256, 512, 512, 1, 1]]
    for c in cases:
        c = c + [1, 0][len(c) - 7:] if len(c) < 9 else c
        for b0 in (0, 4):
            a = c[:8] + [b0]
            # 383 producer stamps per block: 4-chunk tiles 7 per block, 8-chunk tiles 3, 16-chunk tiles 1 -> skip the block's first tile where it fits
            nch = a[3] // 32
            a[7] = 1 if nch <= 16 else 0
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [str(x) for x in a], capture_output=True, text=True)
            print(f"== N={a[0]} {a[1]}x{a[2]} {a[3]}->{a[4]} res={a[5]} gn={a[6]} skip={a[7]} blocks {a[8] * 256}..{a[8] * 256 + 15}", flush=True)
            try:
                analyse(r.stderr, nch)
            except Exception as e:
                print("  analysis failed:", e, r.stderr[-800:])
            if os.environ.get("SDM_TRACE_RAW"):
                print(r.stderr)
