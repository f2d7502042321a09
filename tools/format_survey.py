#!/usr/bin/env python3
"""Which kernel a conversion gets and how fast it is: one source format against many destination formats (and the reverse), 4 HBM-resident
1080p / 4K frames per call.  usage: tools/format_survey.py [same|down|up|same4k]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
FLAGS = int(os.environ.get("SWS_SURVEY_FLAGS", str(SWS_BICUBIC | SWS_BITEXACT)), 0)      # (SWS_BILINEAR = 2, SWS_FAST_BILINEAR = 1, SWS_POINT = 0x10, ...)
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
N = 4
FMTS = ["yuv420p", "yuv422p", "yuv444p", "nv12", "nv21", "p010le", "yuv420p10le", "yuv422p10le", "yuv444p10le", "yuv444p16le", "yuyv422", "uyvy422", "rgb24", "bgr24", "rgba", "bgra",
        "argb", "rgb0", "gbrp", "gbrap", "gbrp10le", "gbrpf32le", "rgb48le", "rgba64le", "rgb565le", "gray8", "gray10le", "gray16le", "yuva420p", "ya8", "x2rgb10le", "ayuv", "vuya", "y210le", "xv30le", "p016le"]
geo = {"same": (1920, 1080, 1920, 1080), "down": (3840, 2160, 1920, 1080), "up": (1280, 720, 1920, 1080), "same4k": (3840, 2160, 3840, 2160)}[mode]
if mode == "same4k": N = 8
sw, sh, dw, dh = geo
if os.environ.get("SWS_SURVEY_GEOM"): sw, sh, dw, dh = [int(v) for v in os.environ["SWS_SURVEY_GEOM"].split(",")]
rows = []
BASES = os.environ.get("SWS_SURVEY_BASES", "yuv420p,nv12,bgra,yuv420p10le").split(",")
for base in BASES:
    for other in FMTS:
        for sf, df in ((base, other), (other, base)):
            if sf == df and mode in ("same", "same4k"): continue
            try:
                ctx = SwsContext(sw, sh, sf, dw, dh, df, FLAGS)
            except Exception as e:
                continue
            hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
            for a, b in zip(hs.planes, src.planes): a[:] = b
            srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
            torch.cuda.synchronize()
            best = 1e9
            try:
                for rep in range(3):
                    t0 = time.perf_counter()
                    for k in range(2): ctx.scale_frames(srcs, dsts)
                    ctx.sync()
                    dt = (time.perf_counter() - t0) / 2 / N * 1e3
                    if rep: best = min(best, dt)
            except Exception as e:
                best = -1
            rows.append((best, f"{sf} -> {df}", ctx.path()))
            ctx.close(); del srcs, dsts
seen = set()
print(f"| conversion ({sw}x{sh} -> {dw}x{dh}, {N} frames per call) | path | ms / frame |")
print("|---|---|---|")
for best, name, path in sorted(rows, reverse=True):
    if name in seen: continue
    seen.add(name)
    print(f"| {name} | {path} | {best:.4f} |")
