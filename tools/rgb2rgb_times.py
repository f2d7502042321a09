#!/usr/bin/env python3
"""Throughput of scaled RGB -> RGB conversions (screen capture / render resize): tools/rgb2rgb_times.py  (options through SWSOPT_<NAME>=v)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT
N = int(os.environ.get("SWS_SHAPES_N", "16"))
OPTS = {k[7:].lower(): int(v) for k, v in os.environ.items() if k.startswith("SWSOPT_")}
CASES = [("bgra",3840,2160,"bgra",1920,1080,SWS_BICUBIC),("rgb24",1920,1080,"rgb24",1280,720,SWS_BICUBIC),("bgra",1920,1080,"bgra",1280,720,SWS_BILINEAR),
         ("bgra",1920,1080,"bgra",3840,2160,SWS_BICUBIC),("rgba",2560,1440,"rgb24",1920,1080,SWS_LANCZOS),("yuv444p",3840,2160,"bgra",1920,1080,SWS_BICUBIC),
         ("yuv420p",3840,2160,"gbrp",1920,1080,SWS_BICUBIC)]
print("| conversion | path / kernel | ms / frame | GB/s (src + dst bytes) |")
print("|---|---|---|---|")
for sf,sw,sh,df,dw,dh,fl in CASES:
    ctx = SwsContext(sw, sh, sf, dw, dh, df, fl | SWS_BITEXACT)
    for k, v in OPTS.items(): ctx.set_option(k, v)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
    torch.cuda.synchronize()
    nbytes = sum(rb * rows for rb, rows in OL.plane_layout(sf, sw, sh)) + sum(rb * rows for rb, rows in OL.plane_layout(df, dw, dh))
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(3): ctx.scale_frames(srcs, dsts)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3 / N * 1e3
        if rep: best = min(best, dt)
    print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {ctx.path()} / {ctx.kernel_name()} | {best:.4f} | {nbytes/best/1e6:.0f} |")
    ctx.close()
