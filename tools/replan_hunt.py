#!/usr/bin/env python3
"""Planner state hygiene, on the CPU box (dry_plan): a context planned under one set of launch options and RE-planned under another must end up with the plan a fresh
context gets under the second set.  A DeviceState field left over from the first plan that a kernel launch of the second reads would make a context wrong for its lifetime
while a fresh one is right -- the forensics of DESIGN.md 8.  usage: python tools/replan_hunt.py <N> <seed>"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SWS_RANDOM_N", "1")
import test_gpu_random as R  # noqa: E402
from librempeg_amd import swscale as S  # noqa: E402

OPTS = ["strip_min_w", "strip_cols_l", "strip_cols_c", "strip_rgb_cols", "no_mixed", "no_wave", "no_march", "no_rgbsrc", "no_strip", "no_strip_dma", "no_dot2", "no_tile",
        "no_strip_short", "no_strip_dma8", "no_strip_rgb2rgb", "no_strip_rgbsrc", "no_rgbsrc2", "no_striprgb_direct", "no_fast_banks", "no_short_forms", "no_strip_range",
        "no_strip_wide", "no_strip_u16", "no_wide_epilogue", "no_generic_kinds", "no_rgbread_kinds", "no_layout_stream", "strip_cols_auto"]
VALUES = {"strip_min_w": [0, 64, 320], "strip_cols_l": [2, 4], "strip_cols_c": [1, 2], "strip_rgb_cols": [2, 4], "strip_cols_auto": [0, 1], "no_wide_epilogue": [0, 1, 2]}


def draw_tune(rng):
    t = {}
    for _ in range(rng.randint(0, 4)):
        k = rng.choice(OPTS)
        t[k] = rng.choice(VALUES.get(k, [0, 1]))
    return t


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    L = S.load_library()
    L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    rng = random.Random(seed)
    cases = R._strip_cases(n, seed) + R._strip_cases(n, seed + 1, R.R4_SRC, R.R4_DST) + R._cases(n, seed + 2)
    bad = total = 0

    def digests(ctx):
        dg = (C.c_uint64 * 3)()
        r = L.sws_hip_plan(ctx.c, dg)
        return (r, ctx.path(), ctx.kernel_name(), dg[1], dg[2])          # (dg[0], the table blocks, may keep a block of the first plan the second does not use)

    for c in cases:
        sw, sh, sf, dw, dh, df, flags = c[:7]
        t1, t2 = draw_tune(rng), draw_tune(rng)
        try:
            a = S.SwsContext(sw, sh, sf, dw, dh, df, flags)
            b = S.SwsContext(sw, sh, sf, dw, dh, df, flags)
        except Exception:
            continue
        total += 1
        a.set_option("dry_plan", 1); b.set_option("dry_plan", 1)
        for k, v in t1.items():
            a.set_option(k, v)
        digests(a)                                   # first plan, under t1
        for k in t1:
            a.set_option(k, {"strip_min_w": 320, "strip_cols_l": 4, "strip_cols_c": 2, "strip_rgb_cols": 4, "strip_cols_auto": 1}.get(k, 0))     # back to the defaults ...
        for k, v in t2.items():
            a.set_option(k, v); b.set_option(k, v)  # ... then t2 on both
        da, db = digests(a), digests(b)
        if da != db:
            bad += 1
            if bad <= 40:
                print("RE-PLAN DIFFERS", c[:7], "first", t1, "then", t2, "\n   re-planned", da, "\n   fresh     ", db, flush=True)
        a.close(); b.close()
    print(f"{total} conversions re-planned, {bad} differ from a fresh context's plan")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
