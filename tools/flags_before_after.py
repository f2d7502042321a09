#!/usr/bin/env python3
"""The round-5 routes behind the flags players and capture tools pass, against the same library with them switched off (no_fast_banks / no_short_forms / no_wave):
ms per frame, 16 HBM-resident frames per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_FAST_BILINEAR, SWS_BITEXACT
N = 16
FB, BL, BC = SWS_FAST_BILINEAR, SWS_BILINEAR, SWS_BICUBIC
CASES = [("yuv420p", 3840, 2160, "yuv420p", 1920, 1080, FB, "no_fast_banks"), ("yuv420p", 1920, 1080, "yuv420p", 1280, 720, FB, "no_fast_banks"), ("yuv420p", 1280, 720, "yuv420p", 1920, 1080, FB, "no_fast_banks"),
         ("yuv420p", 3840, 2160, "bgra", 1920, 1080, FB, "no_fast_banks"), ("nv12", 1920, 1080, "bgra", 1280, 720, FB, "no_fast_banks"), ("yuv422p", 1920, 1080, "yuv420p", 1920, 1080, FB, "no_fast_banks"),
         ("yuyv422", 1920, 1080, "yuv420p", 1280, 720, FB, "no_fast_banks"), ("yuv420p", 1280, 720, "bgra", 1920, 1080, FB, "no_fast_banks"),
         ("yuv420p", 1280, 720, "bgra", 1920, 1080, BL, "no_short_forms"), ("nv12", 1280, 720, "rgb24", 1920, 1080, BL, "no_short_forms"), ("yuv420p", 1920, 1080, "bgra", 3840, 2160, BL, "no_short_forms"),
         ("bgra", 1280, 720, "bgra", 1920, 1080, BL, "no_short_forms"), ("rgb24", 1920, 1080, "rgb24", 3840, 2160, BL, "no_short_forms"), ("yuv444p", 1280, 720, "bgra", 1920, 1080, BL, "no_short_forms"),
         ("yuv420p", 1920, 1080, "bgra", 1280, 1080, BL, "no_short_forms"), ("nv12", 3840, 2160, "bgra", 3840, 2160, BL, "no_short_forms"), ("nv12", 1920, 1080, "rgb24", 1920, 1080, BL, "no_short_forms"),
         ("rgb24", 3840, 2160, "bgra", 1920, 1080, FB, "no_short_forms"), ("bgra", 1366, 768, "yuv420p", 1280, 720, BC, None), ("yuv420p", 3840, 2160, "yuyv422", 3840, 2160, BL, "no_short_forms"), ("nv12", 1920, 1080, "uyvy422", 1920, 1080, BL, "no_short_forms"),
         ("yuv420p", 3840, 2160, "yuyv422", 3840, 2160, BC, "no_wave"), ("nv12", 3840, 2160, "uyvy422", 3840, 2160, BC, "no_wave"), ("bgr24", 3840, 2160, "yuv420p", 3840, 2160, BC, "no_wave"),
         ("gray8", 3840, 2160, "yuv420p", 3840, 2160, BC, "no_mixed"), ("gray16le", 3840, 2160, "bgra", 3840, 2160, BC, "no_wave"), ("gray8", 3840, 2160, "bgra", 1920, 1080, BC, "no_wave"),
         ("x2rgb10le", 3840, 2160, "p010le", 3840, 2160, BC, "no_strip_rgbsrc")]
FL = {FB: "SWS_FAST_BILINEAR", BL: "SWS_BILINEAR", BC: "SWS_BICUBIC"}


def run(sf, sw, sh, df, dw, dh, fl, off):
    ctx = SwsContext(sw, sh, sf, dw, dh, df, fl)
    if off:
        ctx.set_option(off, 1)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for k in range(3): ctx.scale_frames(srcs, dsts)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3 / N * 1e3
        if rep: best = min(best, dt)
    path = ctx.path(); ctx.close()
    return best, path


print("| conversion | flags | option off: path, ms / frame | round 5: path, ms / frame |")
print("|---|---|---|---|")
for sf, sw, sh, df, dw, dh, fl, off in CASES:
    if off is None:
        t1, p1 = run(sf, sw, sh, df, dw, dh, fl, None)
        print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {FL[fl]} | (no switch: width of 4 k + 2, 0.0325 on `main:fused_tile` before) | {p1}, **{t1:.4f}** |")
        continue
    t0, p0 = run(sf, sw, sh, df, dw, dh, fl, off)
    t1, p1 = run(sf, sw, sh, df, dw, dh, fl, None)
    print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {FL[fl]} | `{off}`: {p0}, {t0:.4f} | {p1}, **{t1:.4f}** |")
