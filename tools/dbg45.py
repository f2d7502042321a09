import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from librempeg_amd import SwsContext, DeviceFrame, SWS_BICUBIC, SWS_BITEXACT, SWS_ACCURATE_RND
w, h, n = 3840, 2160, 32
srcs = [DeviceFrame("yuv420p", w, h) for _ in range(n)]
for s in srcs:
    for i in range(3): s.plane_tensor(i).copy_(torch.randint(0, 256, s.plane_tensor(i).shape, dtype=torch.uint8, device="cuda"))
outs = []
for dbg in (0, 4, 5):
    ctx = SwsContext(w, h, "yuv420p", w, h, "rgb24", SWS_BICUBIC | SWS_BITEXACT | SWS_ACCURATE_RND)
    ctx.set_option("debug", dbg)
    dsts = [DeviceFrame("rgb24", w, h) for _ in range(n)]
    torch.cuda.synchronize()
    ctx.set_timing(True)
    r = ctx.scale_frames(srcs, dsts); ctx.sync()
    r = ctx.scale_frames(srcs, dsts); ctx.sync()
    print("debug", dbg, "ret", r, ctx.path(), ctx.kernel_name(), int(dsts[31].plane_tensor(0).sum()), "ms", ctx.last_kernel_ms())
    outs.append(dsts[31].plane_tensor(0).clone())
print(torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]))
