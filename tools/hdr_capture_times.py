import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_BITEXACT
N = 16
CASES = [("x2rgb10le",3840,2160,"p010le",3840,2160),("x2rgb10le",3840,2160,"p010le",1920,1080),("x2rgb10le",1920,1080,"p010le",1920,1080),("x2bgr10le",3840,2160,"yuv420p10le",3840,2160),
         ("x2rgb10le",3840,2160,"nv12",1920,1080),("x2rgb10le",2560,1440,"p010le",2560,1440),("rgb565le",1920,1080,"yuv420p",1920,1080),("bgra",3840,2160,"p010le",3840,2160),
         ("bgra",3840,2160,"yuv420p10le",3840,2160),("rgba64le",3840,2160,"p010le",3840,2160),("gbrp10le",3840,2160,"yuv420p10le",3840,2160),("bgra",3840,2160,"yuv444p",3840,2160),
         ("bgra",1920,1080,"yuv444p10le",1920,1080),("x2rgb10le",3840,2160,"yuv444p10le",3840,2160)]
print("| conversion | path / kernel | ms / frame | GB/s |")
for sf,sw,sh,df,dw,dh in CASES:
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
    torch.cuda.synchronize()
    nbytes = sum(rb * rows for rb, rows in OL.plane_layout(sf, sw, sh)) + sum(rb * rows for rb, rows in OL.plane_layout(df, dw, dh))
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for k in range(3): ctx.scale_frames(srcs, dsts)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3 / N * 1e3
        if rep: best = min(best, dt)
    print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {ctx.path()} / {ctx.kernel_name()} | {best:.4f} | {nbytes/best/1e6:.0f} |")
    ctx.close()
