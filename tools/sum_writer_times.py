#!/usr/bin/env python3
"""Packed destinations behind the strip kernels (path suffix +sum_writer: rgb565 family, x2rgb10, ayuv / vuya / vyu444, y210 / xv30 / xv36) and with the option
no_rgbread_kinds = 2 (the two-pass / single-pass element-per-thread kernels they had before): ms per frame, 4 HBM-resident frames per call, and a byte comparison
of the two results.  usage: tools/sum_writer_times.py [down|up|same]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
mode = sys.argv[1] if len(sys.argv) > 1 else "down"
N = 4
PAIRS = {"down": [(a, b) for a in ("yuv420p", "nv12", "yuv420p10le", "bgra") for b in ("rgb565le", "x2rgb10le", "vuya", "y210le", "xv30le")]}
PAIRS["same"] = PAIRS["down"]
PAIRS["up"] = PAIRS["down"]
geo = {"same": (1920, 1080, 1920, 1080), "down": (3840, 2160, 1920, 1080), "up": (1280, 720, 1920, 1080)}[mode]
sw, sh, dw, dh = geo


def run(sf, df, off):
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT)
    if off:
        ctx.set_option("no_rgbread_kinds", 2)
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


print(f"| conversion ({sw}x{sh} -> {dw}x{dh}, {N} frames per call) | path | element-per-thread kernels, ms / frame | strip kernels + sum writer, ms / frame | x | same bytes |")
print("|---|---|---|---|---|---|")
for sf, df in PAIRS[mode]:
    try:
        t0, path0, o0 = run(sf, df, True)
        t1, path, o1 = run(sf, df, False)
    except Exception as e:
        print(f"| {sf} -> {df} | - | - | - | - | {type(e).__name__} |")
        continue
    print(f"| {sf} -> {df} | {path} | {t0:.4f} | {t1:.4f} | {t0 / t1:.1f} | {'yes' if o0 == o1 else 'NO'} |")
