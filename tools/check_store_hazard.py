"""Scan the engine's gfx950 ISA for a 16-byte BUFFER store whose data registers are rewritten shortly after its issue.

hipcc keeps two wait states between a global / flat store of more than 8 bytes and a VALU write to its data registers, but exempts
`buffer_store_dwordx3/x4` that carry an SGPR offset (the documented rule of earlier parts).  On gfx950 that form showed the hazard
too: a `v_pk_*` two instructions behind the store changed what lanes 12-15 of each 16 wrote (profiles/r03_conv_epilogue_branchfree_ab.txt).
The kernels therefore pin the data registers past the following work (`SDM_PIN_STORE_DATA`, sdm_common.h); this script verifies the
compiled result: for every buffer_store_dwordx4 it reports the distance (in instructions, straight-line) to the first instruction
that writes one of its data registers, and fails when any distance is below MIN_DISTANCE.

usage: python tools/check_store_hazard.py [min_distance=6]      (compiles csrc/sdm_engine.cpp to assembly: ~2 minutes)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "comfyui-sdmatte_amd", "csrc", "sdm_engine.cpp")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.build import CODEGEN_FLAGS      # the product's own code-generation flags
WINDOW = 16


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    min_distance = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "engine.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + CODEGEN_FLAGS + ["--cuda-device-only", "-S", SRC, "-o", out], check=True, capture_output=True)
        lines = open(out).read().split("\n")
    kern, per_kernel = None, {}
    for i, line in enumerate(lines):
        if line.startswith("_Z") and ":" in line:
            kern = line.split(":")[0]
        m = re.match(r"buffer_store_dwordx4\s+(.*)", line.strip())
        if not m:
            continue
        data = regs(m.group(1).split(",")[0].strip())
        n, j, dist = 0, i + 1, None
        while n < WINDOW and j < len(lines):
            u = lines[j].strip()
            j += 1
            if not u or u[0] in ";." or u.endswith(":"):
                continue
            n += 1
            w = re.match(r"(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*|scratch_load\w*)\s+(\S+?),", u)
            if w and not w.group(1).startswith("v_cmp") and regs(w.group(2)) & data:
                dist = n
                break
        per_kernel.setdefault(kern, []).append(dist)
    bad = 0
    for k, v in per_kernel.items():
        near = [x for x in v if x is not None]
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:100]
        print(f"{name}: {len(v)} x buffer_store_dwordx4, data rewritten within {WINDOW} instructions: {len(near)}" + (f" (closest {min(near)})" if near else ""))
        bad += sum(1 for x in near if x < min_distance)
    if bad:
        print(f"FAIL: {bad} store(s) closer than {min_distance} instructions to a write of their data registers")
        sys.exit(1)
    print("ok")


if __name__ == "__main__":
    main()
