import sys; sys.path.insert(0,'/root/repo')
import librempeg_amd as LA
from librempeg_amd import SwsContext
for (sw,sh,sf,dw,dh,df,fl) in [(380,40,"yuv444p",254,22,"bgr24",LA.SWS_BICUBIC|LA.SWS_BITEXACT)]:
    c=SwsContext(sw,sh,sf,dw,dh,df,fl)
    print(sw,sh,sf,dw,dh,df,hex(fl),c.path(),c.kernel_name(), [c.filter(i)[0] for i in range(4)], [c.filter(i)[2][:4] for i in range(4)])
