import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame, SWS_AREA, SWS_BITEXACT
def run(sf, df, sw, sh, dw, dh, tune, seed=5):
    fl = SWS_AREA | SWS_BITEXACT
    o = OL.Oracle(sw, sh, sf, dw, dh, df, fl); p = SwsContext(sw, sh, sf, dw, dh, df, fl)
    for k, v in tune.items(): p.set_option(k, v)
    s = OL.fill_random(OL.Frame(sf, sw, sh), seed); ref = OL.Frame(df, dw, dh, fill=0xA5); o.scale(s, ref)
    hs = HostFrame(sf, sw, sh)
    for a, b in zip(hs.planes, s.planes): a[:] = b
    ds = DeviceFrame(sf, sw, sh).upload(hs); dd = DeviceFrame(df, dw, dh); dd.buf.fill_(0xA5); torch.cuda.synchronize()
    p.scale(ds, dd); p.sync(); out = dd.download()
    for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
        rb = out.row_bytes[i]; d = a[:, :rb] != b[:, :rb]
        if d.any():
            rows = np.nonzero(d.any(axis=1))[0]; cols = np.nonzero(d.any(axis=0))[0]
            print(sf, df, tune, "plane", i, "bad", int(d.sum()), "rows", rows[:20], "cols", cols[:40], "delta", (a[:, :rb].astype(int) - b[:, :rb].astype(int))[d][:20])
        else:
            print(sf, df, tune, "plane", i, "ok", p.path(), p.kernel_name())
T = {"strip_min_w": 0}
run("yuva420p", "yuva420p", 1280, 360, 256, 60, T)
run("gray8", "gray8", 1280, 360, 256, 60, T)
run("yuva444p", "yuva444p", 1280, 360, 256, 60, T)
run("yuva420p", "yuva420p", 1280, 360, 640, 180, T)
run("yuva420p", "yuva420p", 1280, 360, 256, 60, dict(T, strip_cols_auto=0, strip_cols_l=3))
