"""Pins the ORACLE against the reference's fate-pixfmt known answers (tests/ref/pixfmt/*, tests/fate/pixfmt.mak):
53 MD5s of raw frames after <source> -> <fmt> -> <source format> round trips through swscale with
bicubic+accurate_rnd+bitexact.  Between them they exercise the packed-RGB writers (24/32 bpp, X mode), the RGB24/BGR24/
BGRA readers, nv12/p010 readers+writers, 8/10/16-bit planar writers, chroma up/down-scaling in both directions,
MPEG<->JPEG range conversion and the planarCopy depth conversions with and without dither."""
import pytest

import oracle_lib as OL
import fate_patterns as FP


def oracle_convert(src, sfmt, dfmt, dither_none):
    kw = {"dither": 0} if dither_none else {}
    o = OL.Oracle(FP.W, FP.H, sfmt, FP.W, FP.H, dfmt, FP.SWS_FLAGS, **kw)
    dst = OL.Frame(dfmt, FP.W, FP.H)
    assert o.scale(src, dst) == FP.H
    return dst


@pytest.mark.parametrize("key,base,fmt", FP.cases(), ids=[c[0] for c in FP.cases()])
def test_fate_pixfmt_md5(key, base, fmt):
    assert FP.fate_pixfmt_md5(key, base, fmt, oracle_convert) == FP.GOLDEN[key]["md5"]


def test_fate_filter_scalechroma_crc():
    """tests/ref/fate/filter-scalechroma (framecrc, frames 0-1): yuv444p -> yuv420p with -sws_flags +bitexact (bicubic, no
    accurate_rnd) and out_chroma_loc=bottomleft, i.e. dst_h_chr_pos = 0, dst_v_chr_pos = 256 (format.c:554-593): pins the
    bicubic 2:1 chroma down-scale in both directions with non-default chroma siting and the 8-bit planar X writer."""
    import zlib
    for fr, want in zip(FP.vsynth_yuv444_pictures(), FP.SCALECHROMA_CRC):
        o = OL.Oracle(FP.W, FP.H, "yuv444p", FP.W, FP.H, "yuv420p", OL.SWS_BICUBIC | OL.SWS_BITEXACT, dst_h_chr_pos=0, dst_v_chr_pos=256)
        dst = OL.Frame("yuv420p", FP.W, FP.H)
        assert o.scale(fr, dst) == FP.H
        assert zlib.adler32(dst.visible(), 0) & 0xFFFFFFFF == want
