import sys; sys.path.insert(0,'/root/repo')
import librempeg_amd as LA
from librempeg_amd import SwsContext
for (sw,sh,sf,dw,dh,df,fl) in [(3840,2160,"yuv420p",1920,1080,"rgb24",LA.SWS_BICUBIC|LA.SWS_BITEXACT),(96,64,"yuv420p",64,48,"rgb24",LA.SWS_BICUBIC|LA.SWS_BITEXACT),(3840,2160,"yuv420p",1920,1080,"bgra",LA.SWS_BICUBIC),(1920,1080,"yuv420p",1280,720,"rgb24",LA.SWS_BILINEAR),(1280,720,"yuv420p",1920,1080,"bgra",LA.SWS_LANCZOS)]:
    c=SwsContext(sw,sh,sf,dw,dh,df,fl)
    print(sw,sh,sf,dw,dh,df,hex(fl),c.path(),c.kernel_name(), [c.filter(i)[0] for i in range(4)])
