"""CPU tests of sws_frame_setup() / sws_is_noop() / sws_test_frame() on sws_alloc_context()ed ("dynamic") contexts: what the
library derives from the AVFrame fields (libswscale/swscale.c:1503-1619, format.c:305-478, :554-592, graph.c:558-661) is host-side
work, so the conversion it configures can be compared with the oracle's explicit construction without a GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import swscale as S
import frame_props as FP

P = LA.PIX_FMT
BX = LA.SWS_BITEXACT
EINVAL, ENOTSUP = -22, -95


def view(fmt, w, h, **props):
    v = S.SwsFrameView()
    v.width, v.height, v.format = w, h, P[fmt]
    return S.apply_props(v, props)


def dynamic(flags=LA.SWS_BICUBIC | BX, **fields):
    p = LA.SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True)
    f = p.fields()
    f.flags = flags
    for k, v in fields.items():
        setattr(f, k, v)
    return p


def tables_match(p, o, rgb_out):
    for which in range(4):
        a, b = o.filter(which), p.filter(which)
        assert a[0] == b[0], which
        if a[0]:
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), which
    r2y, y2r, co, of, act = p.tables()
    assert r2y == o.rgb2yuv()
    if rgb_out:
        assert y2r == o.yuv2rgb_coeffs()
    oc, oo, oa = o.range_consts()
    assert act == oa and (not act or (co == oc and of == oo))


CASES = [
    # source properties, destination properties: ranges, matrices and chroma siting all come from the frames
    ("yuv420p", dict(color_range="jpeg", colorspace="bt709", chroma_location="left"), "rgb24", {}),
    ("yuv420p", dict(color_range="mpeg", colorspace="bt470bg", chroma_location="topleft"), "bgra", {}),
    ("yuv420p", dict(color_range="mpeg"), "yuv420p", dict(color_range="jpeg")),
    ("yuv420p", dict(color_range="jpeg", chroma_location="left"), "yuv422p", dict(color_range="mpeg", chroma_location="topleft")),
    ("yuvj420p", dict(), "yuv444p", dict(color_range="mpeg")),
    ("rgb24", dict(), "yuv420p", dict(color_range="mpeg", colorspace="bt709", chroma_location="left")),
    ("rgb24", dict(), "nv12", dict(color_range="jpeg", colorspace="smpte240m")),
    ("yuv420p10le", dict(color_range="mpeg", colorspace="bt2020nc", chroma_location="topleft"), "p010le", dict(color_range="mpeg", colorspace="bt2020nc", chroma_location="topleft")),
    ("gray8", dict(), "yuv420p", dict(color_range="mpeg")),
    ("yuv444p", dict(color_range="mpeg", chroma_location="left"), "rgb24", {}),      # no subsampling: the siting is stripped
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[2]}-" + "_".join(f"{v}" for v in list(c[1].values()) + list(c[3].values())))
def test_frame_properties_configure_the_conversion(hiplib, case):
    sfmt, sp, dfmt, dp = case
    sw, sh, dw, dh = 96, 64, 64, 48
    p = dynamic()
    sv, dv = view(sfmt, sw, sh, **sp), view(dfmt, dw, dh, **dp)
    assert hiplib.sws_frame_setup(p.c, C.byref(dv), C.byref(sv)) == 0
    opts, cs = FP.legacy_config(sfmt, sp, dfmt, dp)
    o = OL.Oracle(sw, sh, sfmt, dw, dh, dfmt, LA.SWS_BICUBIC | BX, **opts)
    o.set_colorspace(*cs)
    tables_match(p, o, dfmt in ("rgb24", "bgra"))
    # the context's public fields stay what the caller made them: the conversion lives in a child (graph.c:580-596)
    f = p.fields()
    assert (f.src_w, f.src_h, f.dst_w, f.dst_h, f.src_range, f.dst_range) == (0, 0, 0, 0, 0, 0)
    p.close()


def test_legacy_chroma_position_fields_override_the_frames(hiplib):
    """graph.c:430-444 legacy_chr_pos: the deprecated context fields win over the frame's chroma_location"""
    sw, sh, dw, dh = 96, 64, 64, 48
    p = dynamic(src_h_chr_pos=64, src_v_chr_pos=192)
    sv, dv = view("yuv420p", sw, sh, chroma_location="topleft"), view("yuv422p", dw, dh)
    assert hiplib.sws_frame_setup(p.c, C.byref(dv), C.byref(sv)) == 0
    over = dict(src_h_chr_pos=64, src_v_chr_pos=192)
    opts, cs = FP.legacy_config("yuv420p", dict(chroma_location="topleft"), "yuv422p", {}, overrides=over)
    assert (opts["src_h_chr_pos"], opts["src_v_chr_pos"], opts["dst_v_chr_pos"]) == (64, 192, -513)
    o = OL.Oracle(sw, sh, "yuv420p", dw, dh, "yuv422p", LA.SWS_BICUBIC | BX, **opts)
    o.set_colorspace(*cs)
    tables_match(p, o, False)
    p.close()


def test_interlaced_frames_get_one_conversion_per_field(hiplib):
    """format.c:383-387 + :574-588: half-height fields, the bottom field's chroma sits one luma row lower"""
    sw, sh, dw, dh = 96, 63, 64, 47          # odd heights: the top field has the extra row
    p = dynamic()
    fl = S.AV_FRAME_FLAG_INTERLACED
    sv, dv = view("yuv420p", sw, sh, flags=fl, chroma_location="left"), view("yuv420p", dw, dh, flags=fl, chroma_location="left")
    assert hiplib.sws_frame_setup(p.c, C.byref(dv), C.byref(sv)) == 0
    opts, cs = FP.legacy_config("yuv420p", dict(chroma_location="left"), "yuv420p", dict(chroma_location="left"), True, 0)
    assert opts["src_v_chr_pos"] == 64       # 128 >> 1
    o = OL.Oracle(sw, 32, "yuv420p", dw, 24, "yuv420p", LA.SWS_BICUBIC | BX, **opts)
    tables_match(p, o, False)                # introspection answers for the top field
    assert FP.legacy_config("yuv420p", dict(chroma_location="left"), "yuv420p", {}, True, 1)[0]["src_v_chr_pos"] == (128 + 256) >> 1
    # interlaced -> progressive is refused (swscale.c:1547-1551)
    pv = view("yuv420p", dw, dh)
    assert hiplib.sws_frame_setup(p.c, C.byref(pv), C.byref(sv)) == EINVAL
    p.close()


def test_noop_and_refusals(hiplib):
    L = hiplib
    a, b = view("yuv420p", 64, 32, color_range="mpeg"), view("yuv420p", 64, 32, color_range="mpeg")
    assert L.sws_is_noop(C.byref(a), C.byref(b))
    b.color_range = 2
    assert not L.sws_is_noop(C.byref(a), C.byref(b))
    # rgb frames are always full range / no matrix (sanitize_fmt): the tags do not matter
    r0, r1 = view("rgb24", 64, 32, color_range="mpeg", colorspace="bt709"), view("rgb24", 64, 32)
    assert L.sws_is_noop(C.byref(r0), C.byref(r1))
    # chroma siting only matters with subsampled chroma
    y0, y1 = view("yuv444p", 64, 32, chroma_location="left"), view("yuv444p", 64, 32, chroma_location="top")
    assert L.sws_is_noop(C.byref(y0), C.byref(y1))
    p = dynamic()
    a2 = view("yuv420p", 64, 32, color_range="mpeg")
    assert L.sws_frame_setup(p.c, C.byref(a2), C.byref(a)) == 0
    assert L.sws_hip_path_name(p.c) == b"noop:copy"
    assert L.sws_frame_setup(p.c, C.byref(b), C.byref(a)) == 0      # mpeg -> jpeg: a range conversion, not a copy
    assert L.sws_hip_path_name(p.c) != b"noop:copy" and p.tables()[4] == 1    # range conversion active
    # unsupported matrix (format.c:633-647 sws_test_colorspace): ICTCP = 14
    bad = view("yuv420p", 64, 32, colorspace=14)
    assert not L.sws_test_frame(C.byref(bad), 0)
    assert L.sws_frame_setup(p.c, C.byref(b), C.byref(bad)) == ENOTSUP
    # different primaries / transfer: the reference inserts its 3-D LUT pass (graph.c:760-794); not on this path
    hdr = view("yuv420p10le", 64, 32, color_primaries=9, color_trc=16, colorspace="bt2020nc")
    sdr = view("yuv420p", 64, 32, color_primaries=1, color_trc=1, colorspace="bt709")
    assert L.sws_frame_setup(p.c, C.byref(sdr), C.byref(hdr)) == ENOTSUP
    # ... but unspecified tags are inferred from the other side (format.c:487-552): bt709 -> untagged is a plain conversion
    untagged = view("rgb24", 64, 32)
    assert L.sws_frame_setup(p.c, C.byref(untagged), C.byref(sdr)) == 0
    # SWS_STRICT refuses a conversion that had to guess (swscale.c:1579-1583): here the range of the yuv side
    q = dynamic(flags=LA.SWS_BICUBIC | (1 << 11))
    assert L.sws_frame_setup(q.c, C.byref(untagged), C.byref(view("yuv420p", 64, 32))) == EINVAL
    full = view("yuv420p", 64, 32, color_range="mpeg", colorspace="bt709", chroma_location="left", color_primaries=1, color_trc=1)
    tagged = view("rgb24", 64, 32, color_primaries=1, color_trc=1)
    assert L.sws_frame_setup(q.c, C.byref(tagged), C.byref(full)) == 0
    # option validation (swscale.c:1482-1501)
    q.fields().dither = 99
    assert L.sws_frame_setup(q.c, C.byref(tagged), C.byref(full)) == EINVAL
    # the slice API is for initialised contexts only (swscale.c:1310, :1344, :1371)
    assert L.sws_frame_start(p.c, C.byref(b), C.byref(a)) == EINVAL
    assert L.sws_send_slice(p.c, 0, 16) == EINVAL and L.sws_receive_slice(p.c, 0, 16) == EINVAL
    p.close(); q.close()


def _hw_frame(fmt, w, h, devctx, keep):
    fc = S.SwsHWFramesContext()
    fc.device_ctx = C.pointer(devctx)
    fc.format, fc.sw_format, fc.width, fc.height = S.AV_PIX_FMT_HIP, P[fmt], w, h
    ref = S.SwsBufferRef()
    ref.data = C.addressof(fc)
    ref.size = C.sizeof(fc)
    v = S.apply_props(S.SwsFrameView(), {})
    v.width, v.height, v.format = w, h, S.AV_PIX_FMT_HIP
    v.hw_frames_ctx = C.pointer(ref)
    v.data[0] = 0x1000          # "already allocated" (never dereferenced by sws_frame_setup)
    keep += [fc, ref]
    return v


def test_hardware_frame_checks(hiplib):
    """swscale.c:1515-1538 with HIP in Vulkan's place"""
    L = hiplib
    keep = []
    hip = S.AVHIPDeviceContext(0, None)
    dev = S.SwsHWDeviceContext(None, S.AV_HWDEVICE_TYPE_HIP, C.addressof(hip), None, None)
    dev2 = S.SwsHWDeviceContext(None, S.AV_HWDEVICE_TYPE_HIP, C.addressof(hip), None, None)
    cuda = S.SwsHWDeviceContext(None, 2, None, None, None)           # AV_HWDEVICE_TYPE_CUDA
    p = dynamic()
    s, d = _hw_frame("nv12", 64, 32, dev, keep), _hw_frame("bgra", 64, 32, dev, keep)
    assert L.sws_test_frame(C.byref(s), 0) and L.sws_test_frame(C.byref(d), 1)
    assert L.sws_frame_setup(p.c, C.byref(d), C.byref(s)) == 0
    sw = view("bgra", 64, 32)
    assert L.sws_frame_setup(p.c, C.byref(sw), C.byref(s)) == ENOTSUP          # one hardware frame, one software frame
    d2 = _hw_frame("bgra", 64, 32, dev2, keep)
    assert L.sws_frame_setup(p.c, C.byref(d2), C.byref(s)) == EINVAL           # different devices
    sc, dc = _hw_frame("nv12", 64, 32, cuda, keep), _hw_frame("bgra", 64, 32, cuda, keep)
    assert L.sws_frame_setup(p.c, C.byref(dc), C.byref(sc)) == ENOTSUP         # not a HIP device
    d.data[0] = None
    assert L.sws_frame_setup(p.c, C.byref(d), C.byref(s)) == EINVAL            # hardware frames must be allocated
    bare = view("nv12", 64, 32)
    bare.format = S.AV_PIX_FMT_HIP
    assert not L.sws_test_frame(C.byref(bare), 0)                               # AV_PIX_FMT_HIP without a frames context
    # an initialised context takes hardware frames of its own formats
    q = LA.SwsContext(64, 32, "nv12", 64, 32, "bgra", LA.SWS_BICUBIC)
    d.data[0] = 0x1000
    assert L.sws_frame_setup(q.c, C.byref(d), C.byref(s)) == 0
    assert L.sws_frame_setup(q.c, C.byref(s), C.byref(d)) == EINVAL
    p.close(); q.close()


def test_avframe_mirror_layout():
    """offsets of struct AVFrame (libavutil/frame.h:472-828) measured with the reference's header on x86-64"""
    want = {"data": 0, "linesize": 64, "extended_data": 96, "width": 104, "height": 108, "format": 116, "pict_type": 120,
            "sample_aspect_ratio": 124, "pts": 136, "time_base": 152, "quality": 160, "opaque": 168, "repeat_pict": 176, "buf": 184,
            "extended_buf": 248, "side_data": 264, "nb_side_data": 272, "flags": 276, "color_range": 280, "color_primaries": 284,
            "color_trc": 288, "colorspace": 292, "chroma_location": 296, "best_effort_timestamp": 304, "metadata": 312,
            "decode_error_flags": 320, "hw_frames_ctx": 328, "opaque_ref": 336, "crop_top": 344, "private_ref": 376, "ch_layout": 384,
            "duration": 408, "alpha_mode": 416}
    for k, off in want.items():
        assert getattr(S.SwsFrameView, k).offset == off, k
    assert C.sizeof(S.SwsFrameView) == 424 and C.sizeof(S.SwsFrameSideData) == 40
    assert C.sizeof(S.SwsHWFramesContext) == 80 and S.SwsHWFramesContext.sw_format.offset == 64 and S.SwsHWFramesContext.device_ctx.offset == 16
    assert C.sizeof(S.SwsHWDeviceContext) == 40 and S.SwsHWDeviceContext.hwctx.offset == 16
