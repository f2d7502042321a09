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


# tests/ref/fate/filter-scale-fast-bilinear-wide-edge: framecrc of ONE pixel (3 bytes)
WIDE_EDGE_CRC = 0x0297019b


def wide_edge_source():
    """`color=c=red:s=40000x1,format=yuv444p` (tests/fate/filter-video.mak:191-192): ff_draw_color's limited-range BT.601 red
    (libavfilter/drawutils.c:172-219): Y = 0.299 * 219 + 16 -> 81, U = -0.168736 * 224 + 128 -> 90, V = 0.5 * 224 + 128 -> 240."""
    s = OL.Frame("yuv444p", 40000, 1)
    for pl, v in zip(s.planes, (81, 90, 240)):
        pl[:] = v
    return s


def test_fate_filter_scale_fast_bilinear_wide_edge_crc():
    """scale=40032:1:flags=fast_bilinear,crop=1:1:40031:0 -> adler32 of the last pixel: ff_hyscale_fast_c / ff_hcscale_fast_c with
    their right-edge fix-up at a width where i * xInc no longer fits 31 bits (hscale_fast_bilinear.c)."""
    import zlib
    o = OL.Oracle(40000, 1, "yuv444p", 40032, 1, "yuv444p", OL.SWS_FAST_BILINEAR)
    dst = OL.Frame("yuv444p", 40032, 1)
    assert o.scale(wide_edge_source(), dst) == 1
    px = bytes(int(p[0, 40031]) for p in dst.planes)
    assert zlib.adler32(px, 0) & 0xFFFFFFFF == WIDE_EDGE_CRC
