"""CPU box, no GPU: the planner's answer for 10 000 conversions against a committed table (tests/golden/planner_table.txt.xz).

Every line is `key path kernel table-digest params-digest state-digest` from a dry_plan context (sws_hip_plan(): the planner runs as on a GPU, its table
blocks get fixed fake addresses, uploads are hashed instead of copied).  A changed rule shows up as changed lines; an INTENDED change is
recorded by regenerating the table (SWS_PLANNER_REGEN=1 python -m pytest tests/test_planner_table.py) and reviewing the diff it prints."""
import ctypes as C
import lzma
import os

import pytest

import planner_cases as PC
from librempeg_amd import swscale as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "planner_table.txt.xz")


@pytest.fixture(scope="module")
def lines():
    L = S.load_library()
    L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    return [PC.plan_line(L, *c) for c in PC.cases()]


def test_planner_table(lines):
    if os.environ.get("SWS_PLANNER_REGEN"):
        old = lzma.open(GOLDEN, "rt").read().splitlines() if os.path.exists(GOLDEN) else []
        with lzma.open(GOLDEN, "wt", preset=9) as f:
            f.write("\n".join(lines) + "\n")
        om = {l.split(" ", 1)[0]: l for l in old}
        changed = [l for l in lines if om.get(l.split(" ", 1)[0]) != l]
        print(f"\nplanner table regenerated: {len(lines)} lines, {len(changed)} changed")
        for l in changed[:40]:
            print("  was:", om.get(l.split(" ", 1)[0]), "\n  now:", l)
        return
    want = lzma.open(GOLDEN, "rt").read().splitlines()
    assert len(want) == len(lines), f"{len(lines)} cases, table holds {len(want)}: regenerate (SWS_PLANNER_REGEN=1) after changing tests/planner_cases.py"
    bad = [(w, g) for w, g in zip(want, lines) if w != g]
    msg = "\n".join(f"  table: {w}\n  now:   {g}" for w, g in bad[:25])
    assert not bad, f"{len(bad)} of {len(lines)} plans differ from tests/golden/planner_table.txt.xz (first 25):\n{msg}"


def test_table_covers_every_path_family(lines):
    """the table is only a net if the conversions reach the planner's families: every one of these path names must occur"""
    paths = {l.split(" ")[1] for l in lines if len(l.split(" ")) >= 6}
    for must in ("unscaled:yuv2rgb", "main:fused_rgb_unity", "main:strip_march", "main:strip_rgb2rgb", "main:strip_rgbsrc", "main:strip_packed422",
                 "main:plane1+strip_chroma", "main:two_pass", "main:fused_tile"):
        assert any(p == must or p.startswith(must) for p in paths), f"no conversion of tests/planner_cases.py plans as {must}: {sorted(paths)[:60]}"


def test_dry_plan_context_refuses_to_convert():
    import numpy as np
    ctx = S.SwsContext(64, 36, "yuv420p", 32, 18, "yuv420p", S.SWS_BICUBIC)
    assert ctx.set_option("dry_plan", 1) == 0
    src, dst = S.HostFrame("yuv420p", 64, 36), S.HostFrame("yuv420p", 32, 18)
    for a in src.planes:
        a[:] = 7
    assert ctx.scale(src, dst) < 0
    ctx.close()


def test_plan_and_debug_entry_points_without_a_gpu():
    """sws_hip_plan() / sws_hip_debug_check() / the round-6 option names on planner-only contexts: defined answers, no HIP call"""
    L = S.load_library()
    L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    ctx = S.SwsContext(1280, 720, "yuv420p", 640, 360, "yuv420p", S.SWS_BILINEAR | S.SWS_BITEXACT)
    for name in ("dry_plan", "rccl_tables", "exp0", "exp7"):
        assert ctx.set_option(name, 1 if name == "dry_plan" else 0) == 0, name
    with pytest.raises(ValueError):
        ctx.set_option("no_such_option", 1)
    dg = (C.c_uint64 * 3)()
    assert L.sws_hip_plan(ctx.c, dg) == 0 and dg[0] and dg[1] and dg[2]
    assert ctx.path() == "main:strip_march" and ctx.kernel_name() == "sws_k_strip_dma8"      # BASELINE configs[0]
    again = (C.c_uint64 * 3)()
    assert L.sws_hip_plan(ctx.c, again) == 0 and list(again) == list(dg)                     # planning is idempotent
    n, text = ctx.debug_check()
    assert n == 0, text                                                                       # (nothing on a device to compare: no anomaly, no HIP call)
    assert L.sws_hip_plan(None, dg) < 0
    ctx.close()
