"""The build-time guard of the F8 conv kernels' asynchronous loads (tools/check_async_loads.py, run by comfyui-sdmatte_amd/build.py on the device assembly
of every build): on a synthetic six-step producer loop it must accept a load whose destination is first read six barriers later and reject a register copy
placed right behind the load (what hipcc produced in round 5 when one staged vector had two load sites) and a destination that is overwritten in flight."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_async_loads as G  # noqa: E402


def _asm(step_extra):
    """two kernels with the names the tool looks for; per step: six LDS-DMAs, two asynchronous loads into v[10+4k : 13+4k] and v[50+4k : 53+4k], the hand-over of the vector
    loaded six steps ago (a v_mov from those registers) in front of the load, plus `step_extra(k)` lines behind the load"""
    out = []
    for key in G.KERNELS:
        out.append(key + ":")
        out.append("\ts_barrier")
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


def test_guard_fails_loudly_when_the_loop_is_not_found():
    checked, problems = G.check("_Zsomething_else:\n\ts_endpgm\n")
    assert checked == 0 and len(problems) == len(G.KERNELS)
