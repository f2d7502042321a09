import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import oracle_lib as OL
from librempeg_amd import *
sw, sh, sfmt, dw, dh, dfmt, flags = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7], 0)
o = OL.Oracle(sw, sh, sfmt, dw, dh, dfmt, flags); p = SwsContext(sw, sh, sfmt, dw, dh, dfmt, flags)
src = OL.fill_random(OL.Frame(sfmt, sw, sh), 1); ref = OL.Frame(dfmt, dw, dh, fill=0xA5); o.scale(src, ref)
hs = HostFrame(sfmt, sw, sh)
for a, b in zip(hs.planes, src.planes): a[:] = b
ds = DeviceFrame(sfmt, sw, sh).upload(hs); dd = DeviceFrame(dfmt, dw, dh); dd.buf.fill_(0xA5); torch.cuda.synchronize()
print("ret", p.scale(ds, dd), p.path(), o.path()); p.sync(); out = dd.download()
for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
    rb = out.row_bytes[i]; A = a[:, :rb]; B = b[:, :rb]
    bad = np.argwhere(A != B); print("plane", i, "bad", len(bad), "of", A.size)
    if len(bad):
        y = bad[0][0]; print("row", y, "cols", sorted(set(bad[bad[:,0]==y][:,1]))[:40])
        print("got ", A[y, :48].tolist()); print("want", B[y, :48].tolist())
        for q in range(len(src.planes)): print("src plane", q, src.planes[q][y if q == 0 else y // 2, :24].tolist())
