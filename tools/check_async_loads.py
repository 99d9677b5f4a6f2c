"""Static check of the compiled F8 3x3 conv kernels (k_conv.h, ConvCfg::R4): the producer waves load the activations of the chunk after next through
INLINE-ASM buffer loads whose results arrive up to three steps later, in registers that are carried around the chunk loop.  For hipcc such a result
exists as soon as the asm statement has executed, so nothing stops it from copying or reusing those registers early (a register shuffle at a loop edge,
a second load site merged by a v_mov - seen in round 5: garbage in the tile after).  This tool reads the device assembly of the build (hipcc
-save-temps; comfyui-sdmatte_amd/build.py runs it on every build) and checks, for every such load inside the six-step producer loop, that no instruction
reads or overwrites a destination register before six step barriers have passed - i.e. before the hand-over (take_vec) of the next chunk.
usage: python tools/check_async_loads.py <device .s file>      (exit status 1 on a violation or when the loop cannot be found)"""
import re
import sys

KERNELS = ("_Z16conv_mfma_kernelILi9ELi1ELi8ELi32ELi128ELi32ELi2ELi2ELi1ELi0ELi1ELi1ELi1ELi1ELi1EEv10ConvParams",
           "_Z16conv_mfma_kernelILi9ELi1ELi8ELi32ELi128ELi32ELi2ELi2ELi1ELi0ELi0ELi1ELi1ELi1ELi1EEv10ConvParams")


def _regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _rw(t):
    """(registers read, registers written) of one instruction - conservative: an unknown form counts every register operand as read"""
    ops = t.replace(",", " ").split()
    name, args = ops[0], ops[1:]
    allr = set()
    for a in args:
        allr |= _regs(a)
    if name.startswith(("ds_write", "buffer_store", "global_store", "scratch_store", "s_")) or (name.startswith("buffer_load") and " lds" in t):
        return allr, set()
    w = _regs(args[0]) if args else set()
    r = set()
    for a in args[1:]:
        r |= _regs(a)
    if name.startswith("v_mfma") or "op_sel" in t or name.endswith("_sdwa") or name.startswith(("v_fmac", "v_mac", "v_cvt_pk_bf8", "v_cvt_pk_fp8")):
        r |= w                          # accumulating / partially writing forms read their destination
    return r, w


def check(asm_text, verbose=False):
    lines = asm_text.split("\n")
    problems, checked = [], 0
    for key in KERNELS:
        st = [i for i, l in enumerate(lines) if l.startswith(key + ":")]
        en = [i for i, l in enumerate(lines) if ".amdhsa_kernel " + key in l]
        if not st or not en:
            problems.append(f"{key}: kernel not found in the assembly")
            continue
        K = [l.split(";")[0].strip() for l in lines[st[0]:en[0]]]
        K = [l for l in K if l and not l.startswith(".") and not l.endswith(":")]
        bars = [i for i, l in enumerate(K) if l.startswith("s_barrier")]
        # the producer loop: the first six barrier-delimited segments that each hold LDS-DMAs (buffer_load ... lds) behind the prologue's barrier
        seg_has_dma = [any(K[j].startswith("buffer_load") and " lds" in K[j] for j in range(bars[n] + 1, bars[n + 1])) for n in range(len(bars) - 1)]
        start = next((n for n in range(len(seg_has_dma) - 5) if all(seg_has_dma[n:n + 6])), None)
        if start is None:
            problems.append(f"{key}: six-step producer loop not found (kernel restructured? update tools/check_async_loads.py)")
            continue
        lo, hi = bars[start], bars[start + 6]
        body = list(range(lo + 1, hi + 1))
        n = len(body)
        loads = [(idx, i) for idx, i in enumerate(body) if K[i].startswith("buffer_load_dwordx4") and " lds" not in K[i]]
        if len(loads) < 12:
            problems.append(f"{key}: only {len(loads)} asynchronous loads found in the producer loop")
        for idx, i in loads:
            R = _regs(K[i].split()[1].rstrip(","))
            nb, verdict = 0, None
            for s in range(1, 2 * n):
                t = K[body[(idx + s) % n]]
                if t.startswith("s_barrier"):
                    nb += 1
                r, w = _rw(t)
                if r & R:
                    verdict = ("read", nb, t)
                    break
                if w & R:
                    verdict = ("overwritten", nb, t)
                    break
            checked += 1
            if verdict is None or verdict[0] != "read" or verdict[1] < 6:
                problems.append(f"{key[:60]}...: `{K[i][:60]}` -> {verdict}")
            elif verbose:
                print(f"ok: {K[i][:56]:56s} first read after {verdict[1]} barriers: {verdict[2][:50]}")
    return checked, problems


if __name__ == "__main__":
    checked, problems = check(open(sys.argv[1]).read(), verbose="-v" in sys.argv)
    for p in problems:
        print("VIOLATION:", p)
    print(f"{checked} asynchronous loads checked, {len(problems)} problem(s)")
    sys.exit(1 if problems else 0)
