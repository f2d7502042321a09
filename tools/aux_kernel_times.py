#!/usr/bin/env python3
"""Wall time per frame (HBM-resident frames, stream-synchronised, best of several runs) of the conversions whose kernels are not on a
BASELINE shape: error diffusion, bayer demosaicking, the palette path, float inputs.  Prints a markdown table (profiles/r02_aux_kernels.md)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT

CASES = [
    ("yuv420p 1920x1080 -> rgb8 1920x1080, error diffusion (inner rgb24 pass + sws_k_ed_rgb8)", "yuv420p", 1920, 1080, "rgb8", 1920, 1080, dict(dither=3)),
    ("yuv420p 3840x2160 -> rgb8 1920x1080, error diffusion", "yuv420p", 3840, 2160, "rgb8", 1920, 1080, dict(dither=3)),
    ("yuv420p 1920x1080 -> rgb8, ordered dither (unscaled sws_k_yuv2rgb8_unscaled)", "yuv420p", 1920, 1080, "rgb8", 1920, 1080, {}),
    ("yuv420p 1920x1080 -> monob, error diffusion (sws_k_ed_mono)", "yuv420p", 1920, 1080, "monob", 1920, 1080, dict(dither=3)),
    ("bayer_rggb8 3840x2160 -> rgb24 (sws_k_bayer)", "bayer_rggb8", 3840, 2160, "rgb24", 3840, 2160, {}),
    ("bayer_rggb16le 3840x2160 -> yuv420p (sws_k_bayer, yv12 form)", "bayer_rggb16le", 3840, 2160, "yuv420p", 3840, 2160, {}),
    ("bayer_rggb16le 3840x2160 -> yuv420p10le 1920x1080 (cascade over rgb48)", "bayer_rggb16le", 3840, 2160, "yuv420p10le", 1920, 1080, {}),
    ("pal8 3840x2160 -> bgra (sws_k_update_palette + sws_k_pal2rgb)", "pal8", 3840, 2160, "bgra", 3840, 2160, {}),
    ("pal8 3840x2160 -> yuv420p 1920x1080 (palToY / palToUV readers)", "pal8", 3840, 2160, "yuv420p", 1920, 1080, {}),
    ("rgbaf16le 3840x2160 -> yuv420p10le (half-float reader)", "rgbaf16le", 3840, 2160, "yuv420p10le", 3840, 2160, {}),
]

print("| conversion | path | ms / frame | output Mpix/s |")
print("|---|---|---|---|")
for name, sf, sw, sh, df, dw, dh, opts in CASES:
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT, **opts)
    hs = HostFrame(sf, sw, sh)
    src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    ds = DeviceFrame(sf, sw, sh).upload(hs)
    dd = DeviceFrame(df, dw, dh)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(6):
        t0 = time.perf_counter()
        for k in range(4):
            ctx.scale(ds, dd)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 4 * 1e3
        if rep:
            best = min(best, dt)
    print(f"| {name} | {ctx.path()} | {best:.3f} | {dw * dh / best / 1e3:.0f} |")
    ctx.close()
