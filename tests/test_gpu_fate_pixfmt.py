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


def test_fate_filter_scalechroma_crc_hip():
    """tests/ref/fate/filter-scalechroma through the HIP library (see tests/test_oracle_fate_pixfmt.py)."""
    import zlib
    import oracle_lib as OL
    for fr, want in zip(FP.vsynth_yuv444_pictures(), FP.SCALECHROMA_CRC):
        ctx = SwsContext(FP.W, FP.H, "yuv444p", FP.W, FP.H, "yuv420p", OL.SWS_BICUBIC | OL.SWS_BITEXACT, dst_h_chr_pos=0, dst_v_chr_pos=256)
        hs = HostFrame("yuv444p", FP.W, FP.H)
        for a, b in zip(hs.planes, fr.planes):
            a[:] = b
        ds = DeviceFrame("yuv444p", FP.W, FP.H).upload(hs)
        dd = DeviceFrame("yuv420p", FP.W, FP.H)
        torch.cuda.synchronize()
        assert ctx.scale(ds, dd) == FP.H
        ctx.sync()
        assert zlib.adler32(dd.download().visible(), 0) & 0xFFFFFFFF == want


def test_fate_filter_scale_fast_bilinear_wide_edge_crc_hip():
    """tests/ref/fate/filter-scale-fast-bilinear-wide-edge through the HIP library (see tests/test_oracle_fate_pixfmt.py)."""
    import zlib
    import oracle_lib as OL
    import test_oracle_fate_pixfmt as TO
    ctx = SwsContext(40000, 1, "yuv444p", 40032, 1, "yuv444p", OL.SWS_FAST_BILINEAR)
    hs = HostFrame("yuv444p", 40000, 1)
    for a, b in zip(hs.planes, TO.wide_edge_source().planes):
        a[:] = b
    ds = DeviceFrame("yuv444p", 40000, 1).upload(hs)
    dd = DeviceFrame("yuv444p", 40032, 1)
    torch.cuda.synchronize()
    assert ctx.scale(ds, dd) == 1
    ctx.sync()
    out = dd.download()
    px = bytes(int(p[0, 40031]) for p in out.planes)
    assert zlib.adler32(px, 0) & 0xFFFFFFFF == TO.WIDE_EDGE_CRC
