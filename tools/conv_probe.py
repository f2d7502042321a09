#!/usr/bin/env python3
"""Run a few conversions a few times (for rocprofv3 --kernel-trace): tools/conv_probe.py SWxSH DWxDH src:dst [src:dst ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
sw, sh = map(int, sys.argv[1].split("x")); dw, dh = map(int, sys.argv[2].split("x"))
N = 4
for pair in sys.argv[3:]:
    sf, df = pair.split(":")
    ctx = SwsContext(sw, sh, sf, dw, dh, df, SWS_BICUBIC | SWS_BITEXACT)
    hs = HostFrame(sf, sw, sh); src = OL.fill_random(OL.Frame(sf, sw, sh), 1)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    srcs = [DeviceFrame(sf, sw, sh).upload(hs) for _ in range(N)]; dsts = [DeviceFrame(df, dw, dh) for _ in range(N)]
    torch.cuda.synchronize()
    for k in range(6): ctx.scale_frames(srcs, dsts)
    ctx.sync()
    print(pair, ctx.path(), ctx.kernel_name())
    ctx.close()
