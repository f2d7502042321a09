#!/usr/bin/env python3
"""One frame per sws_scale_frames() call (what a real-time pipeline issues): wall time per call and HIP-event kernel time."""
import sys, time, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT
CASES = [("yuv420p",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",1280,720,SWS_BICUBIC),("yuv420p10le",7680,4320,"p010le",3840,2160,SWS_LANCZOS),
         ("yuv420p",3840,2160,"rgb24",3840,2160,SWS_BICUBIC),("nv12",3840,2160,"bgra",1920,1080,SWS_BICUBIC),("yuv420p",3840,2160,"rgb24",1920,1080,SWS_BICUBIC),
         ("bgra",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"nv12",1920,1080,SWS_BICUBIC),("yuv420p",1280,720,"yuv420p",640,360,SWS_BILINEAR)]
opts = {k[7:].lower(): int(v) for k, v in os.environ.items() if k.startswith("SWSOPT_")}
print("| conversion (1 frame per call) | path / kernel | wall us / call | kernel us (HIP events, median) | GB/s of kernel time |")
print("|---|---|---|---|---|")
for sf,sw,sh,df,dw,dh,fl in CASES:
    ctx = SwsContext(sw, sh, sf, dw, dh, df, fl | SWS_BITEXACT)
    for k, v in opts.items(): ctx.set_option(k, v)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    s = [DeviceFrame(sf, sw, sh).upload(hs)]; d = [DeviceFrame(df, dw, dh)]
    torch.cuda.synchronize()
    nbytes = sum(rb * rows for rb, rows in OL.plane_layout(sf, sw, sh)) + sum(rb * rows for rb, rows in OL.plane_layout(df, dw, dh))
    for _ in range(5): ctx.scale_frames(s, d)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50): ctx.scale_frames(s, d)
    ctx.sync()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    ctx.set_timing(True); ks = []
    for _ in range(12):
        ctx.scale_frames(s, d); ctx.sync(); ks.append(ctx.last_kernel_ms() * 1e3)
    km = statistics.median(ks[2:])
    print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {ctx.path()} / {ctx.kernel_name()} | {wall:.1f} | {km:.1f} | {nbytes / km / 1e3:.0f} |")
    ctx.close()
