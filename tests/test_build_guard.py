"""The build-time guard of the F8 conv kernels' asynchronous loads (tools/check_async_loads.py, run by comfyui-sdmatte_amd/build.py on the device assembly
of every build): on a synthetic six-step producer loop it must accept a load whose destination is first read six barriers later and reject a register copy
placed right behind the load (what hipcc produced in round 5 when one staged vector had two load sites) and a destination that is overwritten in flight."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "comfyui-sdmatte_amd"))
import check_async_loads as G  # noqa: E402

# two F8 3x3 instantiations as hipcc mangles them (with / without fused GroupNorm): the checker finds them by pattern
KERNELS = ("_Z16conv_mfma_kernelILi9ELi1ELi8ELi32ELi128ELi32ELi2ELi2ELi1ELi0ELi1ELi1ELi1ELi1ELi1EEv10ConvParams",
           "_Z16conv_mfma_kernelILi9ELi1ELi8ELi32ELi128ELi32ELi2ELi2ELi1ELi0ELi0ELi1ELi1ELi1ELi1EEv10ConvParams")


def _asm(step_extra, exit_extra=(), reenter=True):
    """two kernels with the names hipcc gives the F8 3x3 instantiations; a loop of six steps - per step: six LDS-DMAs, two asynchronous loads into
    v[10+4k : 13+4k] and v[50+4k : 53+4k], the hand-over of the vector loaded six steps ago (a v_mov from those registers) in front of the load, plus
    `step_extra(k)` lines behind the load - closed by a back edge; behind the loop `exit_extra` (the tile boundary) and the re-entry of the next tile"""
    out = []
    for key in KERNELS:
        out.append(key + ":")
        out.append("\ts_barrier")
        out.append(".LBB0_1:")
        for k in range(6):
            lo = 10 + 4 * k
            out.append(f"\tv_mov_b64_e32 v[100:101], v[{lo}:{lo + 1}]")
            out.append(f"\tv_mov_b64_e32 v[102:103], v[{lo + 40}:{lo + 41}]")
            for _ in range(6):
                out.append("\tbuffer_load_dwordx4 v200, s[52:55], s1 offen lds")
            out.append(f"\tbuffer_load_dwordx4 v[{lo}:{lo + 3}], v0, s[76:79], 0 offen offset:0")              # the two halves of the vector
            out.append(f"\tbuffer_load_dwordx4 v[{lo + 40}:{lo + 43}], v0, s[76:79], 0 offen offset:16")
            out += step_extra(k)
            out.append("\ts_waitcnt vmcnt(10)")
            out.append("\ts_barrier")
        out.append("\ts_cbranch_scc1 .LBB0_1")
        out += list(exit_extra)
        if reenter:
            out.append("\ts_branch .LBB0_1")
        out.append("\ts_endpgm")
        out.append("\t.amdhsa_kernel " + key)
    return "\n".join(out)


def test_guard_accepts_the_hand_over_six_barriers_later():
    checked, problems = G.check(_asm(lambda k: ["\tv_add_f32_e32 v150, v151, v152"]))
    assert checked == 24 and not problems, problems


def test_guard_rejects_an_early_copy_and_an_overwrite_in_flight():
    checked, problems = G.check(_asm(lambda k: ["\tv_mov_b64_e32 v[58:59], v[30:31]"] if k == 5 else []))      # copy of the vector just loaded in step 5
    assert problems and any("read" in p for p in problems)
    checked, problems = G.check(_asm(lambda k: ["\tv_rcp_f32_e32 v14, v150"] if k == 3 else []))             # step 1's destination rewritten two steps later
    assert problems and any("overwritten" in p for p in problems)
    # operands with modifiers name registers too
    checked, problems = G.check(_asm(lambda k: ["\tv_add_f32_e64 v150, -v30, |v151|"] if k == 5 else []))
    assert problems and any("read" in p for p in problems)


def test_guard_walks_the_tile_boundary_behind_the_loop_exit():
    """a vector loaded in the loop's last steps is still in flight when the loop exits: the path through the tile epilogue into the next tile's loop must not touch it
    either (the compiler-placed copy the round-5 failure would have been on THIS path stays invisible to a check of the loop body alone)"""
    ok = ["\tv_add_f32_e32 v150, v151, v152", "\ts_barrier"]
    checked, problems = G.check(_asm(lambda k: [], exit_extra=ok))
    assert checked == 24 and not problems, problems
    checked, problems = G.check(_asm(lambda k: [], exit_extra=["\tv_mov_b32_e32 v160, v31"]))          # step 5's vector copied at the tile boundary
    assert problems and any("exit path" in p for p in problems), problems
    checked, problems = G.check(_asm(lambda k: [], exit_extra=["\tv_mov_b32_e32 v160, v11"]))          # step 0's vector: six barriers have passed - fine
    assert not problems, problems


def test_guard_needs_a_back_edge_and_fails_loudly_when_the_loop_is_not_found():
    txt = _asm(lambda k: []).replace("\ts_cbranch_scc1 .LBB0_1\n", "").replace("\ts_branch .LBB0_1\n", "")      # a peeled copy of the body: no loop
    checked, problems = G.check(txt)
    assert checked == 0 and problems and all("back edge" in p for p in problems)
    checked, problems = G.check("_Zsomething_else:\n\ts_endpgm\n")
    assert checked == 0 and problems


def test_guard_finds_the_kernels_by_pattern():
    assert G.kernel_symbols(_asm(lambda k: [])) == list(KERNELS)
    other = KERNELS[0].replace("ILi9ELi1ELi8E", "ILi1ELi1ELi8E")                       # a 1x1 instantiation is not an F8 3x3 kernel
    assert G.kernel_symbols(other + ":\n") == []


def test_guard_exit_walk_takes_both_branch_successors_and_stops_at_a_drained_counter():
    # the copy sits on the TAKEN side of a conditional branch behind the loop exit
    taken = ["\ts_cbranch_vccnz .LBB0_9", "\ts_branch .LBB0_1", ".LBB0_9:", "\tv_mov_b32_e32 v160, v31"]
    checked, problems = G.check(_asm(lambda k: [], exit_extra=taken))
    assert problems and any("exit path" in p for p in problems), problems
    # the same copy behind s_waitcnt vmcnt(0): everything in flight has landed - not a hazard
    drained = ["\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v160, v31"]
    checked, problems = G.check(_asm(lambda k: [], exit_extra=drained))
    assert not problems, problems
    # a counted wait that leaves loads in flight does not clear it
    counted = ["\ts_waitcnt vmcnt(2)", "\tv_mov_b32_e32 v160, v31"]
    checked, problems = G.check(_asm(lambda k: [], exit_extra=counted))
    assert problems


def test_scalar_load_guard():
    """inline-asm scalar loads of the ping-pong attention kernels (SDM_SLOAD_I32): the destination SGPR may not be read or rewritten before an s_waitcnt that drains
    lgkmcnt - along every path (a register shuffle at a block edge would copy a value that has not arrived)"""
    name = "_Z18attn_d64_pp_kernelILi0ELi1ELi1ELi0EEv10AttnParams"

    def asm(between):
        return "\n".join([name + ":", "\ts_load_dword s53, s[54:55], 0x0"] + between + ["\ts_waitcnt lgkmcnt(0)", "\ts_add_i32 s4, s53, 1", "\ts_endpgm", "\t.amdhsa_kernel " + name])
    checked, problems = G.check_scalar_loads(asm(["\tv_add_f32_e32 v1, v2, v3", "\ts_cbranch_vccz .LBB1_2", "\tv_mul_f32_e32 v1, v1, v1", ".LBB1_2:"]))
    assert checked == 1 and not problems, problems
    checked, problems = G.check_scalar_loads(asm(["\ts_mov_b32 s60, s53"]))                       # the shuffle: a copy of a value still in flight
    assert problems
    checked, problems = G.check_scalar_loads(asm(["\ts_cbranch_vccz .LBB1_2", "\ts_branch .LBB1_3", ".LBB1_2:", "\ts_lshl_b32 s53, s53, 2", ".LBB1_3:"]))      # on one side of a branch only
    assert problems
    checked, problems = G.check_scalar_loads(asm(["\ts_waitcnt vmcnt(0)", "\ts_mov_b32 s60, s53"]))      # a vector-memory wait does not cover it
    assert problems
    assert G.check_scalar_loads("_Zother:\n\ts_load_dword s1, s[2:3], 0x0\n\ts_mov_b32 s4, s1\n\ts_endpgm\n\t.amdhsa_kernel _Zother\n") == (0, [])


def test_dma_loop_guard():
    """the ping-pong attention kernels count their LDS-DMAs on vmcnt by hand: any other vector-memory instruction inside such a loop (a spill, a substituted vector load) breaks the counts"""
    name = "_Z18attn_d64_pp_kernelILi0ELi0ELi0ELi0ELi1ELi1EEv10AttnParams"

    def asm(extra):
        return "\n".join([name + ":", ".LBB9_1:", "\tbuffer_load_dwordx4 v1, s[4:7], s8 offen lds", "\tv_exp_f32_e32 v2, v3"] + extra +
                         ["\ts_waitcnt vmcnt(6)", "\ts_barrier", "\ts_cbranch_scc0 .LBB9_1", "\tglobal_store_dwordx4 v[4:5], v[6:9], off", "\ts_endpgm", "\t.amdhsa_kernel " + name])
    assert G.check_dma_loops(asm([])) == (1, [])                                       # (the store behind the loop is none of its business)
    for bad in ("\tscratch_store_dword off, v9, s32", "\tglobal_load_dword v9, v10, s[2:3]", "\tbuffer_load_dwordx4 v[9:12], v13, s[4:7], 0 offen"):
        checked, problems = G.check_dma_loops(asm([bad]))
        assert problems, bad
    checked, problems = G.check_dma_loops(asm([]).replace(" lds", ""))                  # no DMA loop at all: the kernel was restructured
    assert problems
