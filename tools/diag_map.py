"""bug-hunt helper: where do product and oracle differ (rows / component) for one case"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame


def go(sw, sh, sf, dw, dh, df, flags, opts, tune=None):
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    p = SwsContext(sw, sh, sf, dw, dh, df, flags, **opts)
    for k, v in (tune or {}).items():
        p.set_option(k, v)
    src = OL.fill_random(OL.Frame(sf, sw, sh), 7)
    ref = OL.Frame(df, dw, dh, fill=0xA5)
    o.scale(src, ref)
    hs = HostFrame(sf, sw, sh)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    hd = HostFrame(df, dw, dh)
    for a in hd.planes:
        a[:] = 0xA5
    p.scale(hs, hd)
    print(sf, df, sw, sh, dw, dh, hex(flags), opts, p.path())
    a = hd.planes[0][:, :hd.row_bytes[0]].view(np.uint16).astype(np.int64); b = ref.planes[0][:, :hd.row_bytes[0]].view(np.uint16).astype(np.int64)
    a = a.reshape(dh, -1, 4); b = b.reshape(dh, -1, 4)   # y212: Y0 U Y1 V
    for y in range(dh):
        d = a[y] != b[y]
        if d.any():
            cols = np.argwhere(d.any(axis=1)).ravel()
            print(f" row {y}: comp diffs Y0/U/Y1/V = {d.sum(axis=0).tolist()}, cols {cols[:6].tolist()}..{cols[-3:].tolist()} n={len(cols)}; first got {(a[y][cols[0]] >> 4).tolist()} want {(b[y][cols[0]] >> 4).tolist()}; prev-row want {(b[max(y-1,0)][cols[0]] >> 4).tolist()} next-row want {(b[min(y+1,dh-1)][cols[0]] >> 4).tolist()}")


go(1836, 2, "yuv444p", 1512, 39, "y212le", 2 | 0x80000, {'src_v_chr_pos': 256})
go(1836, 3, "yuv444p", 1512, 12, "y212le", 2 | 0x80000, {'src_v_chr_pos': 256})
