import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT
fl = SWS_BICUBIC | SWS_BITEXACT
for (sf, df) in (("yuv422p16le", "yuv420p"), ("yvyu422", "yuv422p12le")):
    src = OL.fill_random(OL.Frame(sf, 96, 64), 5)
    o = OL.Oracle(96, 64, sf, 96, 64, df, fl)
    ref0 = None; gpu0 = None; bad_o = bad_g = 0
    hs = HostFrame(sf, 96, 64)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    for it in range(1500):
        ref = OL.Frame(df, 96, 64, fill=0xA5); o.scale(src, ref)
        r = ref.visible()
        if ref0 is None: ref0 = r
        elif r != ref0: bad_o += 1
        p = SwsContext(96, 64, sf, 96, 64, df, fl)
        ds = DeviceFrame(sf, 96, 64).upload(hs); dd = DeviceFrame(df, 96, 64); dd.buf.fill_(0xA5)
        torch.cuda.synchronize()
        p.scale(ds, dd); p.sync()
        g = dd.download().visible()
        if gpu0 is None: gpu0 = g
        elif g != gpu0: bad_g += 1
        p.close()
    print(sf, df, "oracle nondeterministic:", bad_o, "gpu nondeterministic:", bad_g, "equal:", ref0 == gpu0)
