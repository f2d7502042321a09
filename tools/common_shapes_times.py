#!/usr/bin/env python3
"""Throughput of common scaling shapes (16 HBM-resident frames per sws_scale_frames() call): which kernel takes them and how fast."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT
N = int(os.environ.get("SWS_SHAPES_N", "16"))
OPTS = {k[7:].lower(): int(v) for k, v in os.environ.items() if k.startswith("SWSOPT_")}
CASES = [("yuv420p",1920,1080,"yuv420p",3840,2160,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",1280,720,SWS_BICUBIC),("yuv420p",3840,2160,"nv12",1920,1080,SWS_BICUBIC),
         ("nv12",1920,1080,"nv12",1280,720,SWS_BILINEAR),("yuv420p",1920,1080,"rgb24",1280,720,SWS_BICUBIC),("yuv420p10le",3840,2160,"yuv420p",1920,1080,SWS_LANCZOS),
         ("rgb24",1920,1080,"yuv420p",1280,720,SWS_BICUBIC),("yuv420p",1920,1080,"bgra",3840,2160,SWS_BICUBIC),("rgb24",3840,2160,"yuv420p",3840,2160,SWS_BICUBIC),
         ("bgra",1920,1080,"nv12",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p10le",1920,1080,SWS_BICUBIC),("p010le",3840,2160,"nv12",1920,1080,SWS_BILINEAR),
         ("yuv422p10le",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),("yuv444p",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),
         ("bgra",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("yuyv422",1920,1080,"yuv420p",1280,720,SWS_BICUBIC),("uyvy422",3840,2160,"nv12",1920,1080,SWS_BILINEAR),
         ("nv12",3840,2160,"bgra",1920,1080,SWS_BICUBIC),("yuv420p10le",3840,2160,"bgra",1920,1080,SWS_BICUBIC),("p010le",3840,2160,"bgra",1920,1080,SWS_BICUBIC),("gbrp",1920,1080,"yuv420p",1280,720,SWS_BICUBIC),("nv12",1920,1080,"rgb24",1280,720,SWS_BILINEAR),("yuv420p",1920,1080,"yuyv422",1920,1080,SWS_BICUBIC),("yuv420p",3840,2160,"uyvy422",1920,1080,SWS_BICUBIC)]
if os.environ.get("SWS_SHAPES_SET") == "ladder":     # the lower rungs of an ABR ladder and thumbnails: ratios of 3:1 and more (filters of 13 .. 33 taps)
    CASES = [("yuv420p",3840,2160,"yuv420p",1280,720,SWS_BICUBIC),("yuv420p",3840,2160,"yuv420p",960,540,SWS_BICUBIC),("yuv420p",3840,2160,"yuv420p",640,360,SWS_BICUBIC),
             ("yuv420p",3840,2160,"yuv420p",1280,720,SWS_LANCZOS),("yuv420p",3840,2160,"yuv420p",1920,1080,SWS_LANCZOS),("yuv420p",1920,1080,"yuv420p",640,360,SWS_BICUBIC),
             ("yuv420p",1920,1080,"yuv420p",426,240,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",640,360,SWS_LANCZOS),("yuv420p10le",3840,2160,"yuv420p10le",960,540,SWS_BICUBIC),
             ("nv12",3840,2160,"nv12",960,540,SWS_BICUBIC),("yuv420p",3840,2160,"rgb24",960,540,SWS_BICUBIC),("yuv420p",3840,2160,"rgb24",640,360,SWS_BICUBIC),
             ("yuv420p",1920,1080,"rgb24",320,180,SWS_BICUBIC),("yuv420p",3840,2160,"yuv420p",960,540,SWS_BILINEAR),("rgb24",3840,2160,"yuv420p",960,540,SWS_BICUBIC),
             ("yuv420p",3840,2160,"yuv420p",480,270,SWS_BICUBIC),("yuv420p",3840,2160,"yuv420p",320,180,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",256,144,SWS_LANCZOS),
             ("yuv420p",3840,2160,"rgb24",320,180,SWS_BICUBIC),("yuv420p",1920,1080,"yuv420p",160,90,SWS_BICUBIC),("yuv420p",1920,1080,"rgb24",128,72,SWS_BICUBIC),
             ("yuv420p",1280,720,"yuv420p",160,90,SWS_BICUBIC)]
if os.environ.get("SWS_SHAPES_SET") == "range":      # MPEG <-> JPEG range conversions (lum/chrRange{To,From}Jpeg_c on the h-scaled lines): MJPEG cameras -> encoders, thumbnails -> JPEG
    CASES = [("yuvj420p",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),("yuvj422p",1280,720,"yuv420p",1280,720,SWS_BICUBIC),("yuvj422p",1920,1080,"nv12",1920,1080,SWS_BICUBIC),
             ("yuvj420p",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"yuvj420p",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"yuvj420p",320,180,SWS_BICUBIC),
             ("yuv420p",3840,2160,"yuvj420p",1280,720,SWS_BICUBIC),("yuv420p",1920,1080,"yuvj420p",1280,720,SWS_BILINEAR),("nv12",1920,1080,"yuvj420p",640,360,SWS_BICUBIC),
             ("rgb24",1920,1080,"yuvj420p",1920,1080,SWS_BICUBIC),("bgra",1920,1080,"yuvj420p",1920,1080,SWS_BICUBIC),("bgra",3840,2160,"yuvj420p",1920,1080,SWS_BICUBIC),
             ("rgb24",1920,1080,"yuvj420p",1280,720,SWS_BICUBIC),("yuvj420p",1920,1080,"yuv420p10le",1920,1080,SWS_BICUBIC),("yuv420p10le",3840,2160,"yuvj420p",1920,1080,SWS_BICUBIC),
             ("yuv420p",1920,1080,"gray8",1920,1080,SWS_BICUBIC),("yuv420p",1920,1080,"gray8",640,360,SWS_BICUBIC),("gray8",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),
             ("yuvj420p",1920,1080,"yuvj420p",1280,720,SWS_BICUBIC),("yuvj444p",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),("yuyv422",1280,720,"yuvj420p",1280,720,SWS_BICUBIC)]
if os.environ.get("SWS_SHAPES_SET") == "wide":       # destinations of 16 bits per component (19-bit intermediates): decoded video -> planar float RGB for inference, 16-bit masters
    CASES = [("nv12",1920,1080,"gbrpf32le",640,360,SWS_BILINEAR),("nv12",1920,1080,"gbrpf32le",960,540,SWS_BICUBIC),("yuv420p",3840,2160,"gbrpf32le",1920,1080,SWS_BICUBIC),
             ("yuv420p",1920,1080,"gbrpf32le",1280,720,SWS_BILINEAR),("nv12",3840,2160,"gbrp16le",1920,1080,SWS_BICUBIC),("yuv420p10le",3840,2160,"gbrpf32le",1920,1080,SWS_BICUBIC),
             ("yuv420p",3840,2160,"yuv420p16le",1920,1080,SWS_BICUBIC),("yuv420p10le",3840,2160,"p016le",1920,1080,SWS_LANCZOS),("yuv420p",1920,1080,"yuv444p16le",1280,720,SWS_BICUBIC),
             ("nv12",1920,1080,"p016le",1280,720,SWS_BILINEAR),("yuv420p",1920,1080,"gray16le",960,540,SWS_BICUBIC),("yuv420p",1280,720,"gbrpf32le",1920,1080,SWS_BICUBIC),
             ("nv12",1920,1080,"gbrpf32le",1920,1080,SWS_BICUBIC),("nv12",1920,1080,"gbrpf32le",224,224,SWS_BILINEAR),("nv12",1920,1080,"gbrpf32le",640,640,SWS_BICUBIC)]
if os.environ.get("SWS_SHAPES_SET") == "u16":        # sources with samples of 16 significant bits (16-bit masters, PNG-16 / TIFF / EXR pictures, float tensors) into delivery formats
    CASES = [("yuv420p16le",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("yuv444p16le",3840,2160,"yuv420p10le",1920,1080,SWS_BICUBIC),("p016le",3840,2160,"nv12",1920,1080,SWS_BILINEAR),
             ("rgb48le",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("rgba64le",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),("gbrpf32le",3840,2160,"yuv420p",1920,1080,SWS_BICUBIC),
             ("gbrpf32le",1920,1080,"nv12",1280,720,SWS_BICUBIC),("gbrp16le",3840,2160,"yuv420p10le",1920,1080,SWS_BICUBIC),("gray16le",3840,2160,"gray8",1920,1080,SWS_BICUBIC),
             ("yuv420p16le",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),("rgb48le",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),("gbrpf32le",1920,1080,"yuv420p",1920,1080,SWS_BICUBIC),
             ("yuv420p16le",3840,2160,"yuv420p16le",1920,1080,SWS_LANCZOS),("rgb48le",1280,720,"yuv420p",1920,1080,SWS_BICUBIC)]
print("| conversion | path / kernel | ms / frame | Gpix/s out | GB/s (src + dst bytes) |")
print("|---|---|---|---|---|")
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
    print(f"| {sf} {sw}x{sh} -> {df} {dw}x{dh} | {ctx.path()} / {ctx.kernel_name()} | {best:.4f} | {dw*dh/best/1e6:.1f} | {nbytes/best/1e6:.0f} |")
    ctx.close()
