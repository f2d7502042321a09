"""-m gpu: the PRODUCT against the reference's fate-filter-pixfmts-null / -scale known answers (tests/ref/fate/filter-pixfmts-*: MD5s of NUT
files, 203 pixel formats each): what `-vf scale,format=F[,scale=200:100]` hands the muxer is produced by sws_scale_frame() on an
sws_alloc_context()ed context that carries only the flags -- the dynamic frame mode (csrc/frames.cpp), driven by the frames' (unspecified)
properties exactly as libavfilter/vf_scale.c:779-866 drives it -- from HBM frames; framing in tests/nut_mux.py, recipe in tests/fate_nut.py.
The oracle twin of this file is tests/test_oracle_fate_nut.py."""
import numpy as np
import pytest

import fate_nut as FN
import oracle_lib as OL

pytestmark = pytest.mark.gpu


def product_convert(src, sfmt, dfmt, dw, dh):
    import torch
    from librempeg_amd.swscale import SwsContext, DeviceFrame, HostFrame
    p = SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True)
    p.fields().flags = FN.FLAGS
    hs = HostFrame(sfmt, src.w, src.h)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    if sfmt == "pal8":
        hs.planes[1][0, :1024] = np.frombuffer(FN.systematic_pal_bgr8(), np.uint8)
    ds = DeviceFrame(sfmt, src.w, src.h).upload(hs)
    dd = DeviceFrame(dfmt, dw, dh)
    ds.props, dd.props = {}, {}
    torch.cuda.synchronize()
    assert p.scale_frame(ds, dd) == 0
    p.sync()
    out = dd.download()
    p.close()
    dst = OL.Frame(dfmt, dw, dh)
    for a, b in zip(dst.planes, out.planes):
        a[:] = b
    return dst


CASES = [(t, f) for t in ("null", "scale") for f in sorted(FN.GOLDEN[t])]


@pytest.mark.parametrize("test,fmt", CASES, ids=[f"{t}-{f}" for t, f in CASES])
def test_fate_filter_pixfmts_md5_on_the_gpu(test, fmt):
    assert FN.md5_of(fmt, test, product_convert) == FN.GOLDEN[test][fmt]


@pytest.mark.parametrize("name", ["scale200", "scale500", "crop_scale", "crop_scale_vflip"])
def test_fate_filter_video_filter_md5_on_the_gpu(name):
    """fate-filter-scale200 / -scale500 / -crop_scale / -crop_scale_vflip (tests/fate/filter-video.mak:511-527): five frames of vsynth1 through the
    scale filter's sws_scale_frame() calls -- yuv420p bicubic up- and down-scaling on real pictures (the strip kernels from 320 columns on)"""
    assert FN.video_filter_md5(name, product_convert) == FN.VIDEO_FILTER_MD5[name]
