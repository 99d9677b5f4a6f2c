"""Per-kernel register / scratch / occupancy table of the engine (hipcc -Rpass-analysis=kernel-resource-usage), to catch
spills and silent occupancy drops after kernel edits.  Usage: python tools/kernel_resources.py [filter-substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "comfyui-sdmatte_amd", "csrc", "sdm_engine.cpp")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_sdmatte_amd.build import CODEGEN_FLAGS      # the product's own code-generation flags
def main():
    filt = sys.argv[1:] or ["conv_mfma", "attn", "gemm"]
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + CODEGEN_FLAGS + os.environ.get("SDM_EXTRA_FLAGS","").split() + ["-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", os.path.join(d, "e.o")],
                           capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, n in zip(blocks, dem):
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1)) if re.search(k + r": (\d+)", b) else -1
        n = n.replace("void ", "")
        if any(f in n for f in filt):
            sc, oc, ld = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
            print(f"vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):4d} scratch={sc:4d} occ={oc} sgpr={g('SGPRs'):3d} lds={ld:6d}  {n[:120]}")
if __name__ == "__main__":
    main()
