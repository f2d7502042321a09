"""The reference's fate-pixfmt known answers (tests/ref/pixfmt/*), run through the HIP library's C-ABI on the GPU:
golden parity of the product itself, not only of the oracle."""
import pytest
import torch

import fate_patterns as FP
from librempeg_amd import SwsContext, HostFrame, DeviceFrame

pytestmark = pytest.mark.gpu


def hip_convert(src, sfmt, dfmt, dither_none):
    kw = {"dither": 0} if dither_none else {}
    ctx = SwsContext(FP.W, FP.H, sfmt, FP.W, FP.H, dfmt, FP.SWS_FLAGS, **kw)
    hs = HostFrame(sfmt, FP.W, FP.H)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    ds = DeviceFrame(sfmt, FP.W, FP.H).upload(hs)
    dd = DeviceFrame(dfmt, FP.W, FP.H)
    torch.cuda.synchronize()
    assert ctx.scale(ds, dd) == FP.H
    ctx.sync()
    return dd.download()


@pytest.mark.parametrize("key,base,fmt", FP.cases(), ids=[c[0] for c in FP.cases()])
def test_fate_pixfmt_md5_hip(key, base, fmt):
    assert FP.fate_pixfmt_md5(key, base, fmt, hip_convert) == FP.GOLDEN[key]["md5"]
