"""flake hunt: every case is computed twice by the oracle and twice by the product (fresh contexts); any disagreement is reported"""
import sys, os, random, time
sys.path.insert(0, 'tests')
import numpy as np
import torch
import test_gpu_random as R
import oracle_lib as OL
from librempeg_amd import SwsContext, HostFrame, DeviceFrame
cases = R._opt_cases(3000, 777)
rng = random.Random(int(sys.argv[1]))
order = list(range(3000)); rng.shuffle(order)
t0 = time.time(); n = 0
def oracle(c, src):
    sw, sh, sf, dw, dh, df, flags, kk, opts, cs = c
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    if cs and o.set_colorspace(*cs) < 0: return None
    ref = OL.Frame(df, dw, dh, fill=0xA5)
    o.scale(src, ref)
    return [p.copy() for p in ref.planes]
def product(c, src, devf):
    sw, sh, sf, dw, dh, df, flags, kk, opts, cs = c
    p = SwsContext(sw, sh, sf, dw, dh, df, flags, **opts)
    if cs and p.set_colorspace(*cs) < 0: return None
    hs = HostFrame(sf, sw, sh)
    for a, b in zip(hs.planes, src.planes): a[:] = b
    hd = HostFrame(df, dw, dh)
    for a in hd.planes: a[:] = 0xA5
    if devf:
        ds = DeviceFrame(sf, sw, sh).upload(hs); dd = DeviceFrame(df, dw, dh); dd.buf.fill_(0xA5)
        torch.cuda.synchronize()
        p.scale(ds, dd); p.sync(); out = dd.download(hd)
    else:
        p.scale(hs, hd); out = hd
    return [(a[:, :rb]).copy() for a, rb in zip(out.planes, out.row_bytes)], p.path()
while time.time() - t0 < float(sys.argv[2]):
    for k in order:
        c = cases[k]
        try:
            OL.Oracle(*c[:7], **c[8])
        except Exception:
            continue
        src = OL.fill_random(OL.Frame(c[2], c[0], c[1]), c[7] + 7)
        o1 = oracle(c, src); o2 = oracle(c, src)
        if o1 is None: continue
        p1, path = product(c, src, True); p2, _ = product(c, src, False); p3, _ = product(c, src, True)
        n += 1
        rb = [x.shape[1] for x in p1]
        def eq(a, b): return all(np.array_equal(x[:, :r], y[:, :r]) for x, y, r in zip(a, b, rb))
        if not (eq(o1, o2) and eq(o1, p1) and eq(o1, p2) and eq(o1, p3)):
            print("MISMATCH case", k, c[:7], c[8], c[9], path, "o1==o2", eq(o1, o2), "p1", eq(o1, p1), "p2", eq(o1, p2), "p3", eq(o1, p3), flush=True)
        if time.time() - t0 > float(sys.argv[2]): break
print("done", sys.argv[1], n, flush=True)
