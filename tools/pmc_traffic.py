"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` to HBM bytes per launch of the conv3x3 kernel.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> <size> <precision> > profiles/pmc_traffic.json
(the passes are taken from `bench.py --timed-only`, so every profiled step is the timed workload)
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) so it is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import csv, json, sys, collections


def total(path, counter):
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if "conv_mfma_kernel<9" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp)


f, nf = total(sys.argv[1], "FETCH_SIZE")
w, nw = total(sys.argv[2], "WRITE_SIZE")
out = {"batch_per_gpu": int(sys.argv[3]), "inference_size": int(sys.argv[4]), "precision": sys.argv[5] if len(sys.argv) > 5 else "fp16x3", "conv3x3_launches_profiled": nf,
       "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_raw_per_launch": w / max(nw, 1),
       "conv3x3_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
       "note": "HBM-side bytes per conv3x3 launch = (2*FETCH_SIZE + WRITE_SIZE) KB, averaged over all conv3x3 launches of the run"}
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
