"""Build tests/emu/_build/libsdmatte_emu.so: the engine sources compiled for the HOST against the fiber
emulator (TEST INFRASTRUCTURE ONLY - see hip_emu.h).  Never loaded by the package."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "comfyui-sdmatte_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libsdmatte_emu.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, "sdm_engine.cpp"), os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "hip_emu.h"),
                                                                                          os.path.join(ROOT, "include", "sdmatte.h")]
    h = hashlib.sha256()
    for f in sorted(deps):
        h.update(open(f, "rb").read())
    stamp = h.hexdigest()
    sf = LIB + ".stamp"
    if os.path.exists(LIB) and os.path.exists(sf) and open(sf).read() == stamp:
        return LIB
    cxx = CXX if os.path.exists(CXX) else "clang++"
    cmd = [cxx, "-x", "c++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-mf16c", "-mavx2", "-mfma", "-DSDM_EMU", "-I", HERE, "-I", CSRC,
           "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed"] + srcs + ["-o", LIB, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("emulator build failed")
    open(sf, "w").write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
