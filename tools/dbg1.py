import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from librempeg_amd import *
for tune in ({}, dict(no_fast_banks=1), dict(strip_min_w=0, no_fast_banks=1), dict(no_fast_banks=1, strip_min_w=0)):
    p = SwsContext(400, 66, "yuv420p", 332, 54, "yuv420p", SWS_FAST_BILINEAR | SWS_BITEXACT)
    for k,v in tune.items(): print("set", k, v, p.set_option(k,v))
    hs = HostFrame("yuv420p", 400, 66); hd = HostFrame("yuv420p", 332, 54); p.scale(hs, hd)
    print(tune, p.path(), p.kernel_name())
