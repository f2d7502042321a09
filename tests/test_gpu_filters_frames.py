"""-m gpu parity: SwsFilter (srcFilter) contexts and the sws_frame_start / sws_send_slice / sws_receive_slice API,
HIP path through the C-ABI vs the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import (SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT,
                           SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT, SWS_POINT)
from librempeg_amd import swscale as S

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT

# (srcFmt, sw, sh, dstFmt, dw, dh, flags, (lumaGBlur, chromaGBlur, lumaSharpen, chromaSharpen, chromaHShift, chromaVShift))
FILTER_CASES = [
    ("yuv420p", 96, 64, "yuv420p", 96, 64, SWS_BICUBIC, (1.2, 0.8, 0.0, 0.0, 0.0, 0.0)),      # same size: no unscaled shortcut
    ("yuv420p", 96, 64, "rgb24", 96, 64, SWS_BILINEAR | BX, (0.0, 2.0, 0.0, 0.0, 0.0, 0.0)),   # chroma blur only: yuv2rgb wrapper is bypassed
    ("yuv420p", 96, 64, "rgb24", 64, 40, SWS_BILINEAR, (0.0, 0.0, 0.6, 0.3, 1.0, 1.0)),
    ("yuv444p", 80, 60, "yuv420p", 120, 90, SWS_LANCZOS, (2.0, 2.0, 0.5, 0.0, 0.0, 0.0)),
    ("rgb24", 72, 50, "yuv420p", 100, 30, SWS_BICUBIC | SWS_ACCURATE_RND, (0.7, 1.5, 0.0, 0.4, 2.0, 0.0)),
    ("nv12", 128, 72, "bgra", 128, 72, SWS_BICUBIC | SWS_FULL_CHR_H_INT, (1.0, 1.0, 0.9, 1.5, 0.0, 2.0)),
    ("yuv420p10le", 96, 64, "p010le", 64, 48, SWS_BICUBIC, (3.0, 0.5, 0.0, 0.0, 0.0, 0.0)),
    ("rgba", 64, 48, "yuva420p", 96, 72, SWS_POINT, (1.5, 1.5, 0.0, 0.0, 0.0, 0.0)),
    ("gray8", 64, 48, "gray16le", 64, 48, SWS_BICUBIC, (1.5, 0.0, 0.0, 0.0, 0.0, 0.0)),
]


def _host_copy(fr, fmt, w, h):
    hs = HostFrame(fmt, w, h)
    for a, b in zip(hs.planes, fr.planes):
        a[:] = b
    return hs


def _assert_same(out, ref, what):
    for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
        rb = out.row_bytes[i]
        if not np.array_equal(a[:, :rb], b[:, :rb]):
            bad = np.argwhere(a[:, :rb] != b[:, :rb])
            raise AssertionError(f"{what} plane {i}: {len(bad)} bytes differ, first at {tuple(bad[0])}: "
                                 f"got {a[tuple(bad[0])]} want {b[tuple(bad[0])]}")


@pytest.mark.parametrize("case", FILTER_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}-{c[3]}_{c[4]}x{c[5]}-{c[6]:x}")
def test_src_filter_parity(case):
    """sws_getContext(..., srcFilter, ...) (utils.c:1256-1263, initFilter :820-870): blur / sharpen / chroma shift."""
    import torch
    sf, sw, sh, df, dw, dh, flags, fp = case
    L = S.load_library()
    f = L.sws_getDefaultFilter(*fp, 0)
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, src_filter=S.filter_to_dict(f))
    plain = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    # the legacy constructor takes the filter too
    c = L.sws_getContext(sw, sh, S.PIX_FMT[sf], dw, dh, S.PIX_FMT[df], flags, C.cast(f, C.c_void_p), None, None)
    assert c
    p = SwsContext(sw, sh, sf, dw, dh, df, flags, src_filter=f, threads=1)
    L.sws_freeFilter(f)
    assert o.path() == "main" and p.path().startswith("main")
    src = OL.fill_random(OL.Frame(sf, sw, sh), 77)
    ref = OL.Frame(df, dw, dh)
    ref0 = OL.Frame(df, dw, dh)
    assert o.scale(src, ref) == dh and plain.scale(src, ref0) == dh
    assert any(not np.array_equal(a, b) for a, b in zip(ref.planes, ref0.planes)), "the filter must change the picture"
    hs = _host_copy(src, sf, sw, sh)
    ds = DeviceFrame(sf, sw, sh).upload(hs)
    dd = DeviceFrame(df, dw, dh)
    dd.buf.fill_(0x3C)
    torch.cuda.synchronize()
    assert p.scale(ds, dd) == dh
    p.sync()
    _assert_same(dd.download(), ref, f"{case} (init_context)")
    dd.buf.fill_(0xC3)
    torch.cuda.synchronize()
    sp, ss = ds.ptrs()
    dp, dstr = dd.ptrs()
    assert L.sws_scale(c, sp, ss, 0, sh, dp, dstr) == dh
    L.sws_hip_sync(c)
    _assert_same(dd.download(), ref, f"{case} (getContext)")
    L.sws_freeContext(c)
    p.close()


FRAME_CASES = [
    (96, 80, "yuv420p", 64, 40, "rgb24", SWS_BICUBIC),
    (96, 80, "yuv420p", 96, 80, "rgb24", SWS_BILINEAR),           # unscaled converter: rows come back slice by slice
    (96, 80, "nv12", 144, 120, "yuv420p10le", SWS_LANCZOS),
    (96, 80, "rgba", 96, 80, "bgra", SWS_POINT),
]


@pytest.mark.parametrize("case", FRAME_CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}")
@pytest.mark.parametrize("dynamic", [False, True], ids=["legacy", "dynamic"])
def test_frame_slice_api(case, dynamic):
    """sws_frame_start() + sws_send_slice()* + sws_receive_slice() + sws_frame_end() (swscale.c:1271-1404)."""
    import torch
    sw, sh, sf, dw, dh, df, flags = case
    L = S.load_library()
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    src = OL.fill_random(OL.Frame(sf, sw, sh), 9)
    ref = OL.Frame(df, dw, dh)
    assert o.scale(src, ref) == dh
    if dynamic:
        p = SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True)
        p.fields().flags = flags
    else:
        p = SwsContext(sw, sh, sf, dw, dh, df, flags)
    ds = DeviceFrame(sf, sw, sh).upload(_host_copy(src, sf, sw, sh))
    dd = DeviceFrame(df, dw, dh)
    dd.buf.fill_(0x11)
    torch.cuda.synchronize()
    sv, dv = ds.view(), dd.view()
    assert L.sws_send_slice(p.c, 0, 16) == -22             # no frame in progress
    if dynamic:   # the slice API is for sws_init_context()ed contexts (swscale.c:1310, :1344, :1371); whole frames still work
        assert L.sws_frame_start(p.c, C.byref(dv), C.byref(sv)) == -22
        assert L.sws_frame_setup(p.c, C.byref(dv), C.byref(sv)) == 0
        assert L.sws_scale_frame(p.c, C.byref(dv), C.byref(sv)) == 0
        p.sync()
        _assert_same(dd.download(), ref, str(case))
        p.close()
        return
    assert L.sws_frame_start(p.c, C.byref(dv), C.byref(sv)) == 0
    align = L.sws_receive_slice_alignment(p.c)
    assert align >= 1
    cuts = [(0, 32), (32, 16), (48, sh - 48)]
    for k, (y0, n) in enumerate(cuts):
        assert L.sws_send_slice(p.c, y0, n) == 0          # ff_range_add only (swscale.c:1337-1351)
        assert L.sws_send_slice(p.c, y0, n) == -22        # ... which refuses rows it already has (utils.c:2394-2404)
        want = dh if k == len(cuts) - 1 else -11          # AVERROR(EAGAIN) until the last source rows are in, then the rows of the slice
        assert L.sws_receive_slice(p.c, 0, dh) == want
    assert L.sws_receive_slice(p.c, 0, align) == align    # a destination slice (rows already there)
    L.sws_frame_end(p.c)
    assert L.sws_receive_slice(p.c, 0, dh) == -22
    p.sync()
    _assert_same(dd.download(), ref, str(case))
    # a frame that does not match a legacy context is refused; a dynamic context re-configures itself
    other = DeviceFrame(sf, sw + 16, sh)
    ov = other.view()
    r = L.sws_frame_setup(p.c, C.byref(dv), C.byref(ov))
    assert r == (0 if dynamic else -22)
    p.sync()
    p.close()
