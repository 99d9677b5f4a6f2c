"""Static check of the compiled F8 3x3 conv kernels (k_conv.h, ConvCfg::R4): the producer waves load the activations of the chunk after next through
INLINE-ASM buffer loads whose results arrive up to three steps later, in registers that are carried around the chunk loop - and, for a tile's last
vectors, across the tile boundary into the next tile's first steps.  For hipcc such a result exists as soon as the asm statement has executed, so nothing
stops it from copying or reusing those registers early (a register shuffle at a loop edge, a second load site merged by a v_mov - seen in round 5:
garbage in the tile after).  This module reads the device assembly of a build (hipcc -save-temps; build.py runs it on every build, BEFORE the library is
accepted) and checks, for every such load inside the six-step producer loop, that no instruction reads or overwrites a destination register before six
step barriers have passed - i.e. before the hand-over (take_vec) of the next chunk - along BOTH continuations of the loop: around its back edge, and out
of its exit through the tile epilogue and the re-entry of the next tile (unconditional branches followed, conditional ones fall through and are ALSO
followed when they are the loop's own back edge).
usage: python check_async_loads.py <device .s file> [-v]      (exit status 1 on a violation or when the loop cannot be found)"""
import re
import sys

# every instantiation of conv_mfma_kernel with NTAPS = 9 and F8 = 1 (first / last of the 15 integer template arguments): found by pattern, not by a
# hard-coded mangled name, so that a new tile parameter does not silently take a kernel out of the check
KERNEL_RE = re.compile(r"^(_Z16conv_mfma_kernelILi9E(?:Li-?\d+E){13}Li1EEv10ConvParams):")

_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def kernel_symbols(asm_text):
    return [m.group(1) for m in (KERNEL_RE.match(l) for l in asm_text.split("\n")) if m]


def _regs(tok):
    """vector registers named by one operand token, whatever modifiers it carries (-v1, |v1|, neg(v1), v1 row_shr:1, v[2:3] ...)"""
    out = set()
    for m in _VREG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _rw(t):
    """(registers read, registers written) of one instruction - conservative: an unknown form counts every register operand as read"""
    parts = t.split(None, 1)
    name = parts[0]
    args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
    allr = set()
    for a in args:
        allr |= _regs(a)
    if name.startswith(("ds_write", "buffer_store", "global_store", "scratch_store", "s_")) or (name.startswith("buffer_load") and " lds" in t):
        return allr, set()
    w = _regs(args[0]) if args else set()
    r = set()
    for a in args[1:]:
        r |= _regs(a)
    if name.startswith("v_mfma") or "op_sel" in t or name.endswith(("_sdwa", "_dpp")) or "row_" in t or "quad_perm" in t or \
            name.startswith(("v_fmac", "v_mac", "v_cvt_pk_bf8", "v_cvt_pk_fp8", "v_permlane", "v_swap")):
        r |= w                          # accumulating / partially writing / lane-masked forms read their destination
    return r, w


def _parse(lines):
    """instructions of one kernel + label -> index of the instruction that follows it"""
    K, labels = [], {}
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.startswith("."):
            if l.endswith(":") and l.startswith(".L"):
                labels[l[:-1]] = len(K)
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(K)
            continue
        K.append(l)
    return K, labels


def _branch_target(t):
    p = t.split()
    return p[1] if len(p) > 1 else None


def check(asm_text, verbose=False):
    lines = asm_text.split("\n")
    problems, checked = [], 0
    keys = kernel_symbols(asm_text)
    if not keys:
        return 0, ["no F8 3x3 instantiation of conv_mfma_kernel found in the assembly"]
    for key in keys:
        st = [i for i, l in enumerate(lines) if l.startswith(key + ":")]
        en = [i for i, l in enumerate(lines) if ".amdhsa_kernel " + key in l]
        if not st or not en:
            problems.append(f"{key}: kernel body not found in the assembly")
            continue
        K, labels = _parse(lines[st[0] + 1:en[0]])
        bars = [i for i, l in enumerate(K) if l.startswith("s_barrier")]
        # the producer loop: six consecutive barrier-delimited segments that each hold LDS-DMAs (buffer_load ... lds) AND close with a back edge to (or
        # in front of) their first segment - a peeled copy of the loop body in front of the loop has no such edge
        seg_has_dma = [any(K[j].startswith("buffer_load") and " lds" in K[j] for j in range(bars[n] + 1, bars[n + 1])) for n in range(len(bars) - 1)]
        start, back = None, None
        for n in range(len(seg_has_dma) - 5):
            if not all(seg_has_dma[n:n + 6]):
                continue
            lo, hi = bars[n], bars[n + 6]
            nxt = bars[n + 7] if n + 7 < len(bars) else len(K)
            be = [j for j in range(lo + 1, nxt) if K[j].startswith(("s_cbranch", "s_branch")) and labels.get(_branch_target(K[j]), 1 << 30) <= lo + 1 and j >= hi - 1]
            be = be or [j for j in range(hi, nxt) if K[j].startswith(("s_cbranch", "s_branch")) and labels.get(_branch_target(K[j]), 1 << 30) <= lo + 1]
            if be:
                start, back = n, be[0]
                break
        if start is None:
            problems.append(f"{key}: six-step producer loop with a back edge not found (kernel restructured? update check_async_loads.py)")
            continue
        lo, hi = bars[start], bars[start + 6]
        body = list(range(lo + 1, hi + 1))
        n = len(body)
        loads = [(idx, i) for idx, i in enumerate(body) if K[i].startswith("buffer_load_dwordx4") and " lds" not in K[i]]
        if len(loads) < 12:
            problems.append(f"{key}: only {len(loads)} asynchronous loads found in the producer loop")

        def walk_exit(first, R, nb0):
            """every path out of the loop (depth-first over (instruction, barriers passed); both successors of a conditional branch): a path is safe once six
            barriers have passed (the hand-over), at an s_waitcnt that drains the vector-memory counter (vmcnt(0): whatever was in flight has landed) and at
            s_endpgm; -> the first violation found, or None"""
            seen, stack = set(), [(first, nb0)]
            while stack:
                j, nb = stack.pop()
                while j < len(K):
                    if (j, nb) in seen:
                        break
                    seen.add((j, nb))
                    t = K[j]
                    if t.startswith("s_endpgm") or (t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t)):
                        break
                    if t.startswith("s_barrier"):
                        nb += 1
                        if nb >= 6:
                            break
                    r, w = _rw(t)
                    if r & R:
                        return ("read", nb, t)
                    if w & R:
                        return ("overwritten", nb, t)
                    if t.startswith(("s_branch", "s_cbranch")):
                        tgt = labels.get(_branch_target(t))
                        if tgt is not None:
                            stack.append((tgt, nb))
                        if t.startswith("s_branch"):
                            break
                    j += 1
            return None

        for idx, i in loads:
            R = _regs(K[i].split(None, 1)[1].split(",")[0])
            # (1) around the back edge
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
                problems.append(f"{key[:60]}...: `{K[i][:60]}` -> {verdict} (around the loop)")
                continue
            # (2) out of the loop exit (the tile boundary): the barriers between the load and the end of the body count, the rest must pass on the exit path
            nb_end, early = 0, None
            for s in range(idx + 1, n):
                t = K[body[s]]
                if t.startswith("s_barrier"):
                    nb_end += 1
                r, w = _rw(t)
                if (r | w) & R:
                    early = t
                    break
            v2 = None if early else walk_exit(hi + 1, R, nb_end)
            if v2 is not None and v2[1] < 6:
                problems.append(f"{key[:60]}...: `{K[i][:60]}` -> {v2} (on the loop's exit path, {nb_end} barriers inside the loop)")
            elif verbose:
                print(f"ok: {K[i][:56]:56s} first read after {verdict[1]} barriers: {verdict[2][:40]} | exit path: {v2}")
    return checked, problems


# ---- second guard: scalar loads issued through inline asm (sdm_common.h SDM_SLOAD_I32: the tile-list entries of attn_d64_pp_kernel, which cannot afford a vector load
#      beside its LDS-DMAs).  For hipcc the result exists as soon as the asm statement has executed; nothing but the operand tie of SDM_SLOAD_WAIT keeps a use - or a
#      register shuffle (s_mov) at a block edge - behind the s_waitcnt.  Checked here for EVERY s_load_dword of the kernels named below (the compiler's own loads
#      satisfy the rule by construction): along every path from the load, no instruction may read or overwrite a destination SGPR before an s_waitcnt that drains lgkmcnt.
SCALAR_KERNEL_RE = re.compile(r"^(_Z18attn_d64_pp_kernelI[A-Za-z0-9_]*EEv10AttnParams):")
_SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def _sregs(tok):
    out = set()
    for m in _SREG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_scalar_loads(asm_text, verbose=False):
    lines = asm_text.split("\n")
    problems, checked = [], 0
    keys = [m.group(1) for m in (SCALAR_KERNEL_RE.match(l) for l in lines) if m]
    for key in keys:
        st = [i for i, l in enumerate(lines) if l.startswith(key + ":")]
        en = [i for i, l in enumerate(lines) if ".amdhsa_kernel " + key in l]
        if not st or not en:
            problems.append(f"{key}: kernel body not found in the assembly")
            continue
        K, labels = _parse(lines[st[0] + 1:en[0]])
        for i, t in enumerate(K):
            if not t.startswith(("s_load_dword", "s_buffer_load_dword")):
                continue
            R = _sregs(t.split(None, 1)[1].split(",")[0])
            checked += 1
            seen, stack, bad = set(), [i + 1], None
            while stack and bad is None:
                j = stack.pop()
                while j < len(K):
                    if j in seen:
                        break
                    seen.add(j)
                    u = K[j]
                    if u.startswith("s_endpgm") or (u.startswith("s_waitcnt") and re.search(r"lgkmcnt\(0\)", u)):
                        break
                    if u.startswith("s_waitcnt") and "lgkmcnt" not in u and "vmcnt" not in u and "expcnt" not in u:      # s_waitcnt <imm>: assume it drains
                        break
                    args = u.split(None, 1)[1] if " " in u else ""
                    if _sregs(args) & R:
                        bad = u
                        break
                    if u.startswith(("s_branch", "s_cbranch")):
                        tgt = labels.get(_branch_target(u))
                        if tgt is not None:
                            stack.append(tgt)
                        if u.startswith("s_branch"):
                            break
                    j += 1
            if bad is not None:
                problems.append(f"{key[:70]}: `{t[:60]}` -> `{bad[:60]}` touches the destination before an s_waitcnt lgkmcnt(0)")
            elif verbose:
                print(f"ok: {key[:40]} {t[:60]}")
    return checked, problems


# ---- third guard: the ping-pong attention kernels wait for their LDS-DMAs with COUNTED vmcnt immediates (k_attn.h: "only the two youngest batches stay in flight").
#      Those counts hold only while the DMAs are the loop's ONLY vector-memory instructions: a register spill (scratch_*), a vector load hipcc substitutes for a scalar
#      one, or a store inside the loop would shift them - silently, towards too lenient.  Every loop (backward branch) of those kernels that issues an LDS-DMA may
#      contain no other buffer_ / global_ / scratch_ / flat_ instruction.
def check_dma_loops(asm_text, verbose=False):
    lines = asm_text.split("\n")
    problems, checked = [], 0
    keys = [m.group(1) for m in (SCALAR_KERNEL_RE.match(l) for l in lines) if m]
    for key in keys:
        st = [i for i, l in enumerate(lines) if l.startswith(key + ":")]
        en = [i for i, l in enumerate(lines) if ".amdhsa_kernel " + key in l]
        if not st or not en:
            problems.append(f"{key}: kernel body not found in the assembly")
            continue
        K, labels = _parse(lines[st[0] + 1:en[0]])
        loops = 0
        for j, t in enumerate(K):
            if not t.startswith(("s_branch", "s_cbranch")):
                continue
            tgt = labels.get(_branch_target(t))
            if tgt is None or tgt > j:
                continue
            body = K[tgt:j + 1]
            if not any(u.startswith("buffer_load") and " lds" in u for u in body):
                continue
            loops += 1
            for u in body:
                if u.startswith(("buffer_", "global_", "scratch_", "flat_")) and not (u.startswith("buffer_load") and " lds" in u) and not u.startswith("buffer_inv") and not u.startswith("buffer_wbl2"):
                    problems.append(f"{key[:70]}: `{u[:60]}` inside a loop that waits for its LDS-DMAs with counted vmcnt")
            checked += 1
        abl = re.search(r"kernelILi(\d+)E", key)
        if loops == 0 and not (abl and int(abl.group(1)) & 8):      # (ABL & 8: the bench-only ablations without DMAs)
            problems.append(f"{key[:70]}: no loop with LDS-DMAs found (kernel restructured? update check_async_loads.py)")
    return checked, problems


if __name__ == "__main__":
    checked, problems = check(open(sys.argv[1]).read(), verbose="-v" in sys.argv)
    c3, p3 = check_dma_loops(open(sys.argv[1]).read())
    checked += c3
    problems += p3
    c2, p2 = check_scalar_loads(open(sys.argv[1]).read(), verbose="-v" in sys.argv)
    checked += c2
    problems += p2
    for p in problems:
        print("VIOLATION:", p)
    print(f"{checked} asynchronous loads checked, {len(problems)} problem(s)")
    sys.exit(1 if problems else 0)
