"""Build recipe of the native engine: `hipcc --offload-arch=gfx950` -> csrc/libsdmatte_hip.so (in-tree, so the
built library travels to the GPU box with the repo snapshot).  gfx950 only; no other backend is compiled."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsdmatte_hip.so")
SOURCES = ["sdm_engine.cpp"]
HEADERS = ["sdm_common.h", "k_conv.h", "k_gemm.h", "k_norm.h", "k_attn.h", "k_misc.h", os.path.join("..", "..", "include", "sdmatte.h")]
# -fno-slp-vectorize: hipcc's SLP pass turns adjacent scalar fp32 adds / multiplies into v_pk_* plus the v_mov that assemble the pairs -
# more issue slots than the scalar form, beside MFMAs (measured: -0.5 % step time without it, profiles/r03_no_slp_ab.txt)
# -amdgpu-sched-strategy=max-ilp: the d=512 attention kernel gains 16 % (9.1 -> 7.6 ms / step), the d=64 one 1 %, the conv kernels (whose
# MFMA / fragment-read order is pinned by sched barriers) are unchanged within noise (profiles/r03_sched_strategy_ab.txt)
FLAGS = ["-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-result", "-Wno-unused-value", "-DNDEBUG", "-munsafe-fp-atomics"]
CODEGEN_FLAGS = [f for f in FLAGS if f not in ("-shared", "-fPIC")]      # for the ISA / resource-usage tools (tools/kernel_resources.py, check_store_hazard.py)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the SDMatte engine needs the ROCm toolchain (gfx950)")


def _stamp():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_all(verbose=False, force=False, extra_flags=(), out=None):
    """out: another output path (bench-only variants such as the -DSDM_CONV_TRACE build of tools/conv_trace.py); default: the product library."""
    lib = out or LIB
    stamp_file = lib + ".stamp"
    stamp = _stamp() + " ".join(extra_flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        if verbose:
            print(f"[sdmatte] {lib} is up to date")
        return lib
    # compiled in a scratch directory with -save-temps: the device assembly is a by-product there, and the F8 3x3 kernel's asynchronous (inline-asm)
    # activation loads are checked against it before the library is accepted (check_async_loads.py, part of this package: no instruction may touch a
    # load's destination registers before the hand-over six barriers later - hipcc has no notion of a result that is still in flight).  The checker is
    # imported FIRST: a missing checker fails the build before two minutes of hipcc, not after.
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("sdmatte_check_async_loads", os.path.join(HERE, "check_async_loads.py"))
    check_async_loads = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(check_async_loads)
    with tempfile.TemporaryDirectory(prefix="sdmatte_build_") as tmp:
        tlib = os.path.join(tmp, "lib.so")
        cmd = [_hipcc()] + FLAGS + ["-save-temps"] + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tlib]
        if verbose:
            print("[sdmatte] " + " ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed building libsdmatte_hip.so")
        if verbose and r.stderr.strip():
            print(r.stderr[-4000:])
        # the device assembly behind -save-temps: whatever this ROCm names it, it is the .s file that holds the gfx950 kernels
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f] or [f for f in os.listdir(tmp) if f.endswith(".s") and "host" not in f]
        if not asm:
            raise RuntimeError("device assembly not found behind -save-temps: cannot check the asynchronous loads of the F8 conv kernel")
        with open(os.path.join(tmp, asm[0])) as fh:
            asm_text = fh.read()
            checked, problems = check_async_loads.check(asm_text)
            c2, p2 = check_async_loads.check_scalar_loads(asm_text)      # the ping-pong attention kernels' scalar loads (tile-list entries through inline asm)
            c3, p3 = check_async_loads.check_dma_loops(asm_text)         # ... and nothing but LDS-DMAs on vmcnt inside their loops (counted waits)
            checked, problems = checked + c2 + c3, problems + p2 + p3
        if problems:
            sys.stderr.write("\n".join(problems) + "\n")
            raise RuntimeError("a build-time guard of the asynchronous loads failed (check_async_loads.py): see the lines above")
        if verbose:
            print(f"[sdmatte] {checked} asynchronous loads / DMA loops checked (F8 conv kernels: vector loads; ping-pong attention kernels: scalar loads, LDS-DMA loops)")
        shutil.move(tlib, lib)
    # a kernel whose body the HOST pass rejects (e.g. inline asm that is only valid for gfx950) is dropped without a diagnostic and leaves
    # an undefined stub symbol: load the library once so that this fails here, in the build container, and not on the GPU box
    # (in a child process: the library must not stay mapped in the builder - a later rebuild to the same path would otherwise meet a stale mapping - and a
    # host without a loadable HIP runtime must not fail an otherwise good build: only an undefined SYMBOL is a build error)
    chk = subprocess.run([sys.executable, "-c", "import ctypes, sys; ctypes.CDLL(sys.argv[1])", lib], capture_output=True, text=True)
    if chk.returncode != 0:
        if "undefined symbol" in chk.stderr:
            sys.stderr.write(chk.stderr)
            raise RuntimeError("libsdmatte_hip.so has an undefined symbol: a kernel body was dropped by the host pass")
        if verbose:
            print("[sdmatte] note: the built library could not be loaded on this host (no HIP runtime?): " + chk.stderr.strip().splitlines()[-1])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return lib


if __name__ == "__main__":
    build_all(verbose=True, force="--force" in sys.argv)
