#!/usr/bin/env python3
"""Scaled RGB sources beyond the 8-bit ones (x2rgb10, rgb565 family, planar RGB of 9 - 14 bits) through the per-kind reader pre-pass + strip kernels
(the default) and with the option no_rgbread_kinds = 1 (the tile / two-pass kernels they had before): ms per frame, 4 HBM-resident frames per call, and a
byte comparison of the two results.  usage: tools/rgbread_kinds_times.py [down|up|same]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
mode = sys.argv[1] if len(sys.argv) > 1 else "down"
N = 4
PAIRS = {"down": [(a, b) for a in ("x2rgb10le", "rgb565le", "gbrp10le", "bgr444le", "y210le", "xv30le", "xv36le", "vuya", "vyu444") for b in ("yuv420p", "nv12", "yuv420p10le", "bgra", "rgb24", "yuv444p")]}
PAIRS["same"] = PAIRS["down"]
PAIRS["up"] = PAIRS["down"]
geo = {"same": (1920, 1080, 1920, 1080), "down": (3840, 2160, 1920, 1080), "up": (1280, 720, 1920, 1080)}[mode]
sw, sh, dw, dh = geo


def run(sf, df, off):
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT)
    if off:
        ctx.set_option("no_rgbread_kinds", 1)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for k in range(2): ctx.scale_frames(srcs, dsts)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 2 / N * 1e3
        if rep: best = min(best, dt)
    hd = HostFrame(df, dw, dh); dsts[N - 1].download(hd)
    out = hd.visible()
    path = ctx.path()
    ctx.close()
    return best, path, out


print(f"| conversion ({sw}x{sh} -> {dw}x{dh}, {N} frames per call) | path | tile / two-pass kernels, ms / frame | reader pre-pass + strip kernels, ms / frame | x | same bytes |")
print("|---|---|---|---|---|---|")
for sf, df in PAIRS[mode]:
    try:
        t0, path0, o0 = run(sf, df, True)
        t1, path, o1 = run(sf, df, False)
    except Exception as e:
        print(f"| {sf} -> {df} | - | - | - | - | {type(e).__name__} |")
        continue
    print(f"| {sf} -> {df} | {path} | {t0:.4f} | {t1:.4f} | {t0 / t1:.1f} | {'yes' if o0 == o1 else 'NO'} |")
