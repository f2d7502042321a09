#!/usr/bin/env python3
"""Kernel time (HIP events on the context's stream) of the pure layout / depth converters and of the same-size conversions with a vertical
chroma step, N HBM-resident frames per sws_scale_frames() call; GB/s = visible source + destination bytes, frac = GB/s over 8 TB/s."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
W, H = (3840, 2160)
CASES = [("yuv420p", "yuv420p10le"), ("yuv420p10le", "yuv420p"), ("yuv420p", "nv12"), ("nv12", "yuv420p"), ("yuyv422", "yuv420p"), ("uyvy422", "yuv422p"),
         ("yuv422p", "yuyv422"), ("yuv420p", "yuv420p"), ("yuv444p", "nv24"), ("nv12", "p010le"), ("yuv422p10le", "yuv420p"), ("yuv444p", "yuv420p"),
         ("yuv422p", "yuv420p"), ("gbrp", "yuv420p"), ("rgb24", "yuv444p"), ("yuv444p10le", "yuv420p10le")]
if len(sys.argv) > 1:
    CASES = [tuple(a.split(":")) for a in sys.argv[1:]]
opts = {}
for k in list(os.environ):
    if k.startswith("SWSOPT_"):
        opts[k[7:].lower()] = int(os.environ[k])
print(f"| conversion ({W}x{H}) | path / kernel | frames | kernel us / call (min / median) | GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|")
for sf, df in CASES:
    nbytes = sum(rb * rows for rb, rows in OL.plane_layout(sf, W, H)) + sum(rb * rows for rb, rows in OL.plane_layout(df, W, H))
    N = max(8, int(1.2e9 // nbytes))
    ctx = SwsContext(W, H, sf, W, H, df, SWS_BICUBIC | SWS_BITEXACT)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.set_timing(True)
    hs = HostFrame(sf, W, H); src = OL.fill_random(OL.Frame(sf, W, H), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, W, H).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, W, H) for _ in range(N)]
    torch.cuda.synchronize()
    ts = []
    for rep in range(12):
        ctx.scale_frames(srcs, dsts); ctx.sync()
        if rep >= 2: ts.append(ctx.last_kernel_ms() * 1e3)
    mn, md = min(ts), statistics.median(ts)
    gbs = nbytes * N / (md * 1e-6) / 1e9
    print(f"| {sf} -> {df} | {ctx.path()} / {ctx.kernel_name()} | {N} | {mn:.1f} / {md:.1f} | {gbs:.0f} | {gbs / 8000:.3f} |")
    ctx.close(); del srcs, dsts
    torch.cuda.empty_cache()
