#!/usr/bin/env python3
"""Outputs of 256 .. 1023 columns: the tile kernels (strip_min_w = 1024, the planner's threshold until round 3) against the strip kernels (strip_min_w = SWS_NARROW_ALT,
default 320 = the planner's threshold now), per frame at 1 / 16 / 128 frames per call.  SWS_NARROW_SET=small: outputs below 640 columns."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT
CASES = [("yuv420p",1280,720,"yuv420p",640,360,SWS_BILINEAR),("yuv420p",1280,720,"yuv420p",640,360,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",960,540,SWS_BICUBIC),
         ("yuv420p",1920,1080,"yuv420p",854,480,SWS_BICUBIC),("yuv420p",1280,720,"yuv420p",854,480,SWS_BICUBIC),("nv12",1920,1080,"nv12",960,540,SWS_BILINEAR),
         ("yuv420p10le",1920,1080,"yuv420p10le",960,540,SWS_BICUBIC),("yuv420p",1920,1080,"rgb24",960,540,SWS_BICUBIC),("yuv420p",640,360,"yuv420p",854,480,SWS_BICUBIC),
         ("yuv420p",854,480,"yuv420p",640,360,SWS_LANCZOS)]
ALT = int(os.environ.get("SWS_NARROW_ALT", "320"))
if os.environ.get("SWS_NARROW_SET") == "small":
    CASES = [("yuv420p",960,540,"yuv420p",480,270,SWS_BICUBIC),("yuv420p",854,480,"yuv420p",426,240,SWS_BICUBIC),("yuv420p",640,360,"yuv420p",320,180,SWS_BILINEAR),
             ("yuv420p",1280,720,"yuv420p",512,288,SWS_BICUBIC),("nv12",960,540,"nv12",480,270,SWS_BICUBIC),("yuv420p",512,288,"yuv420p",256,144,SWS_BICUBIC),
             ("yuv420p",320,180,"yuv420p",480,270,SWS_BICUBIC),("yuv420p10le",960,540,"yuv420p",480,270,SWS_LANCZOS),("yuv420p",352,288,"yuv420p",352,240,SWS_BICUBIC)]
print("| conversion | frames / call | strip_min_w = 1024: path | ms / frame | strip_min_w = 320: path | ms / frame |")
print("|---|---|---|---|---|---|")
for sf,sw,sh,df,dw,dh,fl in CASES:
    for N in (1, 16, 128):
        res = []
        for opt in (1024, ALT):
            ctx = SwsContext(sw, sh, sf, dw, dh, df, fl | SWS_BITEXACT)
            if opt: ctx.set_option("strip_min_w", opt)
            hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
            for a, b in zip(hs.planes, src.planes): a[:] = b
            srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(6):
                t0 = time.perf_counter()
                for k in range(5): ctx.scale_frames(srcs, dsts)
                ctx.sync()
                dt = (time.perf_counter() - t0) / 5 / N * 1e3
                if rep: best = min(best, dt)
            res.append((ctx.path(), best)); ctx.close()
        print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {N} | {res[0][0]} | {res[0][1]:.4f} | {res[1][0]} | {res[1][1]:.4f} |")
