#!/usr/bin/env python3
"""The element-per-thread (generic) kernels before / after their per-kind instantiations (k_generic_kinds.hip): ms per frame with the option
no_generic_kinds = 1 (the all-kinds kernels of k_generic.hip) and = 0 (the default), 4 HBM-resident frames per call, and a byte comparison of the
two results.  usage: tools/generic_kinds_times.py [same|down|up]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
N = 4
PAIRS = {
 "same": [("yuv420p", "rgba64le"), ("yuv420p10le", "rgb48le"), ("bgra", "xv30le"), ("yuv420p", "y210le"), ("rgba64le", "yuv420p"), ("bgra", "rgba64le"), ("nv12", "rgb48le"),
          ("yuv420p10le", "rgb565le"), ("rgb48le", "yuv420p10le"), ("yuv420p", "x2rgb10le"), ("y210le", "yuv420p"), ("bgra", "gbrpf32le"), ("gbrpf32le", "bgra"),
          ("rgb565le", "yuv420p"), ("x2rgb10le", "yuv420p10le"), ("yuv444p16le", "bgra"), ("bgra", "vuya"), ("ayuv", "bgra"), ("p016le", "bgra"), ("gbrp10le", "bgra"),
          ("xv30le", "bgra"), ("ya8", "bgra"), ("gray10le", "bgra"), ("bgra", "rgb565le"), ("yuv420p10le", "xv30le"), ("yuv420p", "p016le"), ("gbrp10le", "yuv420p")],
 "down": [("bgra", "y210le"), ("bgra", "p016le"), ("bgra", "rgb565le"), ("gbrpf32le", "bgra"), ("ayuv", "bgra"), ("rgba64le", "bgra"), ("yuv444p16le", "bgra"), ("gbrp10le", "bgra"),
          ("bgra", "rgba64le"), ("gbrap", "bgra"), ("bgra", "vuya"), ("xv30le", "bgra"), ("y210le", "bgra"), ("gbrpf32le", "yuv420p"), ("rgb48le", "bgra"), ("yuv420p10le", "y210le"),
          ("yuv420p10le", "rgba64le"), ("yuv420p", "p016le"), ("yuv420p", "rgb565le"), ("yuv420p", "x2rgb10le"), ("gbrp10le", "yuv420p"), ("gbrp10le", "nv12"), ("rgb565le", "yuv420p"),
          ("x2rgb10le", "yuv420p"), ("y210le", "yuv420p"), ("p016le", "yuv420p")],
}
PAIRS["up"] = PAIRS["down"]
geo = {"same": (1920, 1080, 1920, 1080), "down": (3840, 2160, 1920, 1080), "up": (1280, 720, 1920, 1080)}[mode]
sw, sh, dw, dh = geo


def run(sf, df, off):
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT)
    if off:
        ctx.set_option("no_generic_kinds", 1)
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


print(f"| conversion ({sw}x{sh} -> {dw}x{dh}, {N} frames per call) | path | all-kinds kernels, ms / frame | per-kind kernels, ms / frame | x | same bytes |")
print("|---|---|---|---|---|---|")
for sf, df in PAIRS[mode]:
    try:
        t0, path, o0 = run(sf, df, True)
        t1, path, o1 = run(sf, df, False)
    except Exception as e:
        print(f"| {sf} -> {df} | - | - | - | - | {type(e).__name__} |")
        continue
    print(f"| {sf} -> {df} | {path} | {t0:.4f} | {t1:.4f} | {t0 / t1:.1f} | {'yes' if o0 == o1 else 'NO'} |")
