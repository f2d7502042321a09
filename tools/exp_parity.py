#!/usr/bin/env python3
"""GPU box (or, with SWS_SUITE_ON_EMU=1 and the emulation library, the CPU box): the round-6 experiment instantiations (sws_hip_set_option exp0 .. exp3) against the oracle, bit-exact, before any of them is timed.
usage: python tools/exp_parity.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401  (SWS_SUITE_ON_EMU=1: the same cases against the x86 emulation build on the CPU box, tests/hipemu/README.md)
import test_gpu_parity as T  # noqa: E402
from librempeg_amd import SWS_BILINEAR, SWS_BICUBIC, SWS_LANCZOS, SWS_BITEXACT  # noqa: E402

C3B = [dict(exp1=1), dict(exp0=3, exp1=1, exp2=5), dict(exp0=2, exp1=1, exp2=6), dict(exp0=2, exp2=8),
       dict(strip_cols_l=2, strip_cols_c=1, exp2=7), dict(strip_cols_l=2, strip_cols_c=1, exp2=7, exp1=1),
       dict(strip_cols_l=3, exp4=1), dict(strip_cols_l=3, exp4=1, exp1=1), dict(strip_cols_l=3, exp4=1, no_strip_dma=1)]
C1 = [dict(exp3=1)]
ok = True
for tune in [{}] + C3B:
    for (sw, sh, dw, dh, sf, df, fl) in [(1920, 1080, 960, 540, "yuv420p10le", "p010le", SWS_LANCZOS | SWS_BITEXACT), (2048, 300, 1024, 150, "yuv420p10le", "yuv420p10le", SWS_BICUBIC | SWS_BITEXACT),
                                         (1536, 200, 1000, 77, "yuv422p10le", "p010le", SWS_BILINEAR | SWS_BITEXACT), (3840, 128, 1280, 60, "yuv420p12le", "yuv420p", SWS_LANCZOS | SWS_BITEXACT)]:
        try:
            p = T.run_case(sw, sh, sf, dw, dh, df, fl, seed=5, tune=dict(strip_min_w=0, **tune))
            print("ok  ", tune, sf, df, f"{sw}x{sh}->{dw}x{dh}", p[0], flush=True)
        except AssertionError as e:
            ok = False
            print("FAIL", tune, sf, df, f"{sw}x{sh}->{dw}x{dh}", str(e)[:300], flush=True)
for tune in [{}] + C1:
    for (sw, sh, dw, dh, sf, df, fl) in [(1280, 720, 640, 360, "yuv420p", "yuv420p", SWS_BILINEAR | SWS_BITEXACT), (1280, 722, 640, 362, "yuv422p", "yuv420p", SWS_BILINEAR | SWS_BITEXACT)]:
        for batch in (False,):
            try:
                p = T.run_case(sw, sh, sf, dw, dh, df, fl, seed=6, tune=tune)
                print("ok  ", tune, sf, df, f"{sw}x{sh}->{dw}x{dh}", p[0], flush=True)
            except AssertionError as e:
                ok = False
                print("FAIL", tune, sf, df, f"{sw}x{sh}->{dw}x{dh}", str(e)[:300], flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
