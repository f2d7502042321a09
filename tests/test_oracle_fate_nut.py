"""Pins the ORACLE (and tests/frame_props.py, the restatement of the dynamic mode's property handling) against the reference's
fate-filter-pixfmts-copy / -null / -scale known answers: per pixel format the MD5 of the NUT file FATE's video_filter() writes
(tests/ref/fate/filter-pixfmts-*, 203 formats each; framing restated in tests/nut_mux.py, recipe in tests/fate_nut.py).
-copy and -null hold the same MD5s (both filters pass the converted picture on); -scale adds the 352x288 -> 200x100 resize INSIDE each
format: the only reference goldens for scaled output per format, for the 8 / 4 bpp and 16-bit packed RGB writers, pal8 and the packed YUV families."""
import pytest

import fate_nut as FN
import frame_props as FP
import oracle_lib as OL


def oracle_convert(src, sfmt, dfmt, dw, dh):
    if sfmt == "pal8":                                     # a bgr8 picture read through its systematic palette
        pal = OL.Frame("pal8", src.w, src.h)
        pal.planes[0][:, :src.w] = src.planes[0][:, :src.w]
        import numpy as np
        pal.planes[1][0, :1024] = np.frombuffer(FN.systematic_pal_bgr8(), np.uint8)
        src = pal
    dst = OL.Frame(dfmt, dw, dh)
    return FP.oracle_convert(src, {}, dst, {}, FN.FLAGS)


CASES = [(t, f) for t in ("null", "scale") for f in sorted(FN.GOLDEN[t])]


@pytest.mark.parametrize("test,fmt", CASES, ids=[f"{t}-{f}" for t, f in CASES])
def test_fate_filter_pixfmts_md5(test, fmt):
    assert FN.md5_of(fmt, test, oracle_convert) == FN.GOLDEN[test][fmt]


def test_copy_and_null_hold_the_same_answers():
    assert FN.GOLDEN["copy"] == FN.GOLDEN["null"]


@pytest.mark.parametrize("name", sorted(FN.VIDEO_FILTER_MD5))
def test_fate_filter_video_filter_md5(name):
    """fate-filter-null / -crop / -vflip / -crop_vflip (five frames: the NUT framing alone, several packets), fate-filter-scale200 / -scale500 /
    -crop_scale / -crop_scale_vflip (tests/fate/filter-video.mak:508-527): yuv420p scaled by the scale filter at five geometries"""
    assert FN.video_filter_md5(name, oracle_convert) == FN.VIDEO_FILTER_MD5[name]
