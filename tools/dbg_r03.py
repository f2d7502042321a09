#!/usr/bin/env python3
"""debug: which stage of a composite path differs from the oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import run_case
def t(name, *a, **k):
    try:
        r = run_case(*a, **k); print("OK  ", name, r)
    except AssertionError as e:
        print("FAIL", name, str(e)[:300])
fl = 0xc0004
S = dict(strip_min_w=0)
t("yuyv->yuyv strip", 60, 3, "yuyv422", 302, 44, "yuyv422", fl, seed=30256, tune=S)
t("yuyv->yuyv generic", 60, 3, "yuyv422", 302, 44, "yuyv422", fl, seed=30256, tune=dict(no_mixed=1))
t("yuyv->yuv422p strip", 60, 3, "yuyv422", 302, 44, "yuv422p", fl, seed=30256, tune=S)
t("yuv422p->yuyv strip", 60, 3, "yuv422p", 302, 44, "yuyv422", fl, seed=30256, tune=S)
t("yuv422p->yuv422p strip", 60, 3, "yuv422p", 302, 44, "yuv422p", fl, seed=30256, tune=S)
t("yuv422p->yuv422p default", 60, 3, "yuv422p", 302, 44, "yuv422p", fl, seed=30256)
t("yuv422p->yuv422p strip h=4", 60, 4, "yuv422p", 302, 44, "yuv422p", fl, seed=30256, tune=S)
t("yuv422p->yuv422p strip 60x3->302x3", 60, 3, "yuv422p", 302, 3, "yuv422p", fl, seed=30256, tune=S)
fl2 = 0xc0200
t("c2 strip", 340, 2, "yuv444p12le", 352, 25, "yuv420p10be", fl2, seed=12080, tune=S)
t("c2 strip le", 340, 2, "yuv444p12le", 352, 25, "yuv420p10le", fl2, seed=12080, tune=S)
t("c2 default", 340, 2, "yuv444p12le", 352, 25, "yuv420p10be", fl2, seed=12080)
t("c2 strip 8bit", 340, 2, "yuv444p", 352, 25, "yuv420p", fl2, seed=12080, tune=S)
t("c2 strip h=4", 340, 4, "yuv444p12le", 352, 25, "yuv420p10le", fl2, seed=12080, tune=S)
