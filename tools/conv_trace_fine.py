"""Sub-step timeline of the F8 3x3 kernel's producer wave (bench helper; -DSDM_CONV_TRACE -DSDM_CONV_TRACE2 build named by SDM_TRACE_LIB): per step
k = 0..5 the cycles spent in [LDS writes + register hand-over | DMA issue | load issue | transform (+ high-plane writes at k = 5) | end-of-step wait |
barrier].  usage: SDM_TRACE_LIB=... python tools/conv_trace_fine.py [N H W Cin Cout res gn]"""
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [int(x) for x in sys.argv[1:8]] if len(sys.argv) >= 8 else [4, 1024, 1024, 128, 128, 1, 1]
nch = args[3] // 32
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_trace.py"), "--child"] + [str(x) for x in args + [0, 0]], capture_output=True, text=True)
acc = [[[] for _ in range(6)] for _ in range(6)]
for m in re.finditer(r"\[trace\] block (\d+) producer n=\d+: ([\d ]+)", r.stderr):
    v = [int(x) for x in m.group(2).split()]
    tiles, cur = [], None
    for x in v:
        if x & 1:
            cur = []
            tiles.append(cur)
        if cur is not None:
            cur.append(x & ~1)
    for t in tiles[1:]:                       # skip the block's first tile
        if len(t) != 4 + 36 * nch:
            continue
        for c in range(1, nch):               # chunks 1..
            for k in range(6):
                b = 3 + (c * 6 + k) * 6       # released(prev) = t[b-1]; s1..s4 = t[b..b+3]; arrive = t[b+4]; release = t[b+5]
                pts = [t[b - 1]] + t[b:b + 6]
                for j in range(6):
                    acc[k][j].append(pts[j + 1] - pts[j])
mean = lambda x: sum(x) / max(len(x), 1)
print(f"N={args[0]} {args[1]}x{args[2]} {args[3]}->{args[4]} res={args[5]} gn={args[6]}: producer wave 0, chunks 1.., {len(acc[0][0])} samples per cell")
print("  step   writes+take   DMA issue   load issue   transform   end wait   barrier   total")
for k in range(6):
    row = [mean(acc[k][j]) for j in range(6)]
    print(f"  k={k}   " + "   ".join(f"{x:9.0f}" for x in row) + f"   {sum(row):7.0f}")
if not acc[0][0]:
    print(r.stderr[-1500:])
