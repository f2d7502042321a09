#!/usr/bin/env python3
"""The random conversions of tests/test_gpu_random.py (every generator: sizes 1 .. 2 100, all scalers and flags, options, colourspaces, launch heuristics) through context
construction and the PLANNER only -- no GPU (option dry_plan).  Meant to be run under the sanitizer builds:

    tools/asan_env.sh python tools/plan_hunt.py <N per generator> <seed>

A heap overflow or a read of uninitialised stack in filter / table construction or in the planner for some odd geometry would damage host memory of a long-running process:
the signature of the rare events of DESIGN.md 8.  SWS_PLAN_DUMP=<file> writes every case's answer: two runs -- the shipped library and the `make uninit` build
(automatic variables pre-filled with a pattern) under MALLOC_PERTURB_ -- must give identical files.  Also checks that planning the same conversion twice gives the same digests."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SWS_RANDOM_N", "1")      # (the generators are evaluated at import for pytest's parametrisation: keep that draw tiny)
import test_gpu_random as R  # noqa: E402
from librempeg_amd import swscale as S  # noqa: E402


def plan(L, sw, sh, sf, dw, dh, df, flags, opts=None, cs=None, tune=None):
    try:
        ctx = S.SwsContext(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
    except Exception:
        return None
    try:
        for k, v in (tune or {}).items():
            ctx.set_option(k, v)
        ctx.set_option("dry_plan", 1)
        if cs and ctx.set_colorspace(*cs) < 0:
            return None
        dg = (C.c_uint64 * 3)()
        r = L.sws_hip_plan(ctx.c, dg)
        return (r, ctx.path(), dg[0], dg[1], dg[2])
    finally:
        ctx.close()


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    L = S.load_library()
    L.sws_hip_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    total = refused = errors = unstable = 0
    dump = open(os.environ["SWS_PLAN_DUMP"], "w") if os.environ.get("SWS_PLAN_DUMP") else None     # one line per case: compare two libraries' runs with diff
    gens = [("conversions", R._cases(n, seed)), ("options", R._opt_cases(n, seed + 1)), ("strip family", R._strip_cases(n, seed + 2)),
            ("round-4 routes", R._strip_cases(n, seed + 3, R.R4_SRC, R.R4_DST) if hasattr(R, "R4_SRC") else []),
            ("few rows", R._short_cases(n, seed + 4)), ("batches", R._batch_cases(n, seed + 5)), ("unaligned", R._odd_cases(n, seed + 6)), ("slice sequences", R._slice_cases(n, seed + 7))]
    for name, cases in gens:
        for c in cases:
            sw, sh, sf, dw, dh, df, flags = c[:7]
            opts = next((x for x in c[7:] if isinstance(x, dict) and ("dither" in x or "src_range" in x or "threads" in x)), None)
            tune = next((x for x in c[7:] if isinstance(x, dict) and x is not opts and any(k.startswith(("strip_", "no_")) for k in x)), None)
            cs = next((x for x in c[7:] if isinstance(x, tuple) and len(x) == 7 and all(isinstance(v, int) for v in x)), None)
            a = plan(L, sw, sh, sf, dw, dh, df, flags, opts, cs, tune)
            total += 1
            if dump:
                dump.write(f"{name} {c[:7]} {a}\n")
            if a is None:
                refused += 1
                continue
            if a[0] < 0:
                errors += 1
                print("plan error", a[0], c[:7], flush=True)
                continue
            if total % 7 == 0:          # plan a sample twice: a heap- or stack-dependent plan shows up as a changed digest
                b = plan(L, sw, sh, sf, dw, dh, df, flags, opts, cs, tune)
                if b != a:
                    unstable += 1
                    print("UNSTABLE PLAN", c[:7], a, b, flush=True)
        print(f"{name}: {len(cases)} cases", flush=True)
    print(f"total {total}, refused {refused}, plan errors {errors}, unstable {unstable}")
    return 1 if errors or unstable else 0


if __name__ == "__main__":
    sys.exit(main())
