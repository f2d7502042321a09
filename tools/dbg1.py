import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from librempeg_amd import *
for (w,h) in ((3840,2160),(1920,1080)):
  for fl in (SWS_BILINEAR, SWS_BILINEAR|SWS_BITEXACT, SWS_BICUBIC|SWS_BITEXACT):
    for tune in ({}, dict(no_short_forms=1)):
        p = SwsContext(w,h,"nv12",w,h,"bgra",fl)
        for k,v in tune.items(): p.set_option(k,v)
        hs=HostFrame("nv12",w,h); hd=HostFrame("bgra",w,h); p.scale(hs,hd)
        print(w,h,hex(fl),tune,p.path(),p.kernel_name())
