#!/usr/bin/env python3
"""Wall time per frame of common same-size conversions that take the generic element-per-thread kernels (profiles/r02_aux_kernels.md)."""
import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT, SWS_ACCURATE_RND
CASES = [("rgb24",3840,2160,"yuv420p",3840,2160,0),("bgra",3840,2160,"yuv420p",3840,2160,0),("bgra",3840,2160,"nv12",3840,2160,0),("rgb24",3840,2160,"yuv444p",3840,2160,0),
         ("bgra",3840,2160,"yuv420p",1920,1080,0),("rgb24",1920,1080,"yuv420p",1920,1080,0),("bgr24",3840,2160,"yuv420p",3840,2160,0),("bgr24",3840,2160,"yuv420p",3840,2160,SWS_ACCURATE_RND),
         ("gbrp",3840,2160,"yuv420p",3840,2160,0),("rgb48le",3840,2160,"yuv420p10le",3840,2160,0),("yuv420p",3840,2160,"yuv444p",3840,2160,0),("yuv422p",3840,2160,"yuv420p",3840,2160,0),
         ("yuv420p",3840,2160,"nv12",3840,2160,0),("yuyv422",3840,2160,"yuv420p",3840,2160,0),("yuv420p",3840,2160,"yuyv422",3840,2160,0)]
for sf,sw,sh,df,dw,dh,fl in CASES:
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT | fl)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    ds = DeviceFrame(sf, sw, sh).upload(hs); dd = DeviceFrame(df, dw, dh)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(8): ctx.scale(ds, dd)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 8 * 1e3
        if rep: best = min(best, dt)
    print(f"{sf} {sw}x{sh} -> {df} {dw}x{dh} flags={fl:#x} | {ctx.path()} | {best:.3f} ms | {dw*dh/best/1e3:.0f} Mpix/s")
    ctx.close()
