"""Reduce a rocprofv3 --pmc SQ pass of `bench.py --timed-only` (counters: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
[SQ_INSTS_VALU_MFMA_MOPS_F8] SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) to one row per kernel and to the matrix-pipe
busy fraction of the kernel families `bench.py` reports:

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x chip cycles),   chip cycles = SQ_BUSY_CYCLES / 32 (summed over the 32 shader
                   engines; the same number as GRBM_GUI_ACTIVE / 8 XCDs, printed next to it when present)

(SQ_VALU_MFMA_BUSY_CYCLES counts the cycles the matrix pipe of a SIMD is busy: 32 per v_mfma_f32_32x32x16_f16, 64 per K = 64 fp8 MFMA,
MI355X_MICROARCH.md "Per-instruction cycle constants".)

usage: python tools/pmc_sq.py <counter_collection.csv> <batch> <size> <precision> <out.csv> > profiles/pmc_sq.json"""
import collections
import csv
import json
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
counters = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
    counters.add(r["Counter_Name"])
counters = sorted(counters)


def busy(c):
    cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    return c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc > 0 else None


with open(sys.argv[5], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Dispatches", "mfma_busy_frac", "chip_cycles_per_dispatch(SQ_BUSY/32)", "chip_cycles_per_dispatch(GRBM/8)"] + counters)
    for k in sorted(rows, key=lambda k: -rows[k].get("SQ_BUSY_CYCLES", 0.0)):
        c, n = rows[k], len(disp[k])
        b = busy(c)
        w.writerow([k, n, "" if b is None else f"{b:.4f}", f"{c.get('SQ_BUSY_CYCLES', 0.0) / 32.0 / n:.0f}",
                    f"{c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / n:.0f}" if "GRBM_GUI_ACTIVE" in c else ""] + [c.get(x, 0.0) for x in counters])

FAM = {"conv3x3": ("conv_mfma_kernel<9",), "gemm": ("conv_mfma_kernel<1", "gemm_p3_kernel"), "attn_d64": ("attn_d64_",), "attn_d512": ("attn_d512_kernel",)}
out = {"batch_per_gpu": int(sys.argv[2]), "inference_size": int(sys.argv[3]), "precision": sys.argv[4],
       "note": "matrix-pipe busy fraction per kernel family = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32), summed over every "
               "dispatch of the family in a `bench.py --timed-only` run"}
for name, key in FAM.items():
    agg = collections.defaultdict(float)
    n = 0
    for k, c in rows.items():
        if any(x in k for x in key):
            n += len(disp[k])
            for x, v in c.items():
                agg[x] += v
    b = busy(agg)
    out[f"{name}_dispatches"] = n
    out[f"{name}_mfma_busy_frac"] = None if b is None else round(b, 4)
    if agg.get("SQ_WAVE_CYCLES"):
        out[f"{name}_wait_inst_any_frac_of_wave_cycles"] = round(agg.get("SQ_WAIT_INST_ANY", 0.0) / agg["SQ_WAVE_CYCLES"], 4)
try:      # the sources + flags the profiled library was built from (bench.py shows these numbers only beside the same build)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    load_package()
    from comfyui_sdmatte_amd import build as _B
    out["build_stamp"] = _B._stamp()
except Exception:
    out["build_stamp"] = None
print(json.dumps(out))
