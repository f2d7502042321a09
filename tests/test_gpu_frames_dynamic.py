"""sws_scale_frame() / sws_scale_frames() on sws_alloc_context()ed contexts (libswscale/swscale.c:1404-1480): the conversion comes from
the frames' own properties.  The oracle is configured explicitly with what the reference's graph.c:558-661 derives from those
properties (tests/frame_props.py); the pixels must be identical.  Also: interlaced frames (one conversion per field), the no-op
copy, frames of format AV_PIX_FMT_HIP with a hardware frames context, and a context-free upload / download through the
device-level helpers integration/hwcontext_hip.c is written against."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import swscale as S
from librempeg_amd.swscale import SwsContext, DeviceFrame, HostFrame
import frame_props as FP

pytestmark = pytest.mark.gpu
BX = LA.SWS_BITEXACT
P = LA.PIX_FMT


def dynamic(flags, **fields):
    p = SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True)
    f = p.fields()
    f.flags = flags
    for k, v in fields.items():
        setattr(f, k, v)
    return p


def host_copy(fr):
    hs = HostFrame(fr.fmt, fr.w, fr.h)
    for a, b in zip(hs.planes, fr.planes):
        a[:] = b
    return hs


def assert_same(out, ref, what):
    for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
        rb = out.row_bytes[i]
        if not np.array_equal(a[:, :rb], b[:, :rb]):
            bad = np.argwhere(a[:, :rb] != b[:, :rb])
            raise AssertionError(f"{what} plane {i}: {len(bad)} bytes differ, first at {tuple(bad[0])}: got {a[tuple(bad[0])]} want {b[tuple(bad[0])]}")


CASES = [
    # sw, sh, sfmt, source props, dw, dh, dfmt, destination props, flags, interlaced
    (96, 64, "yuv420p", dict(color_range="jpeg", colorspace="bt709", chroma_location="left"), 64, 48, "rgb24", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "yuv420p", dict(color_range="jpeg", colorspace="bt709", chroma_location="left"), 96, 64, "rgb24", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "yuv420p", dict(color_range="mpeg", colorspace="bt470bg", chroma_location="topleft"), 128, 80, "bgra", {}, LA.SWS_BILINEAR | BX | LA.SWS_ACCURATE_RND, False),
    (96, 64, "yuv420p", dict(color_range="mpeg"), 96, 64, "yuv420p", dict(color_range="jpeg"), LA.SWS_BICUBIC | BX, False),
    (96, 64, "yuv420p", dict(color_range="jpeg", chroma_location="left"), 64, 40, "yuv422p", dict(color_range="mpeg", chroma_location="topleft"), LA.SWS_LANCZOS | BX, False),
    (96, 64, "yuvj420p", {}, 64, 40, "yuv444p", dict(color_range="mpeg"), LA.SWS_BICUBIC | BX, False),
    (96, 64, "rgb24", {}, 64, 40, "yuv420p", dict(color_range="mpeg", colorspace="bt709", chroma_location="left"), LA.SWS_BICUBIC | BX, False),
    (96, 64, "rgb24", {}, 96, 64, "nv12", dict(color_range="jpeg", colorspace="smpte240m"), LA.SWS_BICUBIC | BX | LA.SWS_ACCURATE_RND, False),
    (128, 64, "yuv420p10le", dict(color_range="mpeg", colorspace="bt2020nc", chroma_location="topleft"), 64, 32, "p010le",
     dict(color_range="mpeg", colorspace="bt2020nc", chroma_location="topleft"), LA.SWS_LANCZOS | BX, False),
    (96, 64, "nv12", dict(color_range="mpeg", colorspace="bt709"), 96, 64, "bgra", {}, LA.SWS_BICUBIC | BX, False),
    # the formats added late in round 2: palette and bayer sources are "RGB-like" (sanitize_fmt, format.c:305-311), the 8 bpp destination dithers
    (96, 64, "pal8", {}, 64, 40, "yuv420p", dict(color_range="mpeg", colorspace="bt709"), LA.SWS_BICUBIC | BX, False),
    (96, 64, "pal8", {}, 96, 64, "bgra", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "bayer_grbg16le", {}, 64, 40, "yuv444p", dict(color_range="jpeg", colorspace="bt470bg"), LA.SWS_BILINEAR | BX, False),
    (96, 64, "bayer_bggr8", {}, 96, 64, "rgb24", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "yuv420p", dict(color_range="mpeg", colorspace="bt709", chroma_location="left"), 64, 40, "rgb8", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "yuv444p", dict(color_range="mpeg", colorspace="bt709"), 64, 40, "bgr4_byte", {}, LA.SWS_BICUBIC | BX, False),           # 4:4:4 source: error diffusion
    (96, 64, "rgbaf16le", {}, 64, 40, "yuva420p", dict(color_range="mpeg", colorspace="bt709"), LA.SWS_BICUBIC | BX, False),
    (96, 64, "grayf16le", dict(color_range="jpeg"), 64, 40, "gray8", {}, LA.SWS_BICUBIC | BX, False),
    (96, 64, "uyyvyy411", dict(color_range="mpeg"), 64, 40, "yuv420p", dict(color_range="mpeg"), LA.SWS_BICUBIC | BX, False),
    # interlaced: two half-height conversions on every second row; odd heights give the top field the extra row
    (96, 63, "yuv420p", dict(chroma_location="left", color_range="mpeg"), 64, 47, "yuv420p", dict(chroma_location="left", color_range="mpeg"), LA.SWS_BICUBIC | BX, True),
    (96, 64, "yuv420p", dict(color_range="mpeg", colorspace="bt709"), 96, 64, "rgb24", {}, LA.SWS_BICUBIC | BX | LA.SWS_ACCURATE_RND, True),
    (96, 64, "yuv422p", dict(color_range="mpeg"), 48, 32, "yuv420p", dict(color_range="jpeg", chroma_location="topleft"), LA.SWS_BILINEAR | BX, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[6]}_{c[4]}x{c[5]}-{c[8]:x}" + ("-tff" if c[9] else ""))
def test_frame_properties_drive_the_conversion(case):
    import torch
    sw, sh, sfmt, sp, dw, dh, dfmt, dp, flags, interlaced = case
    src = OL.fill_random(OL.Frame(sfmt, sw, sh), 5)
    ref = FP.oracle_convert(src, sp, OL.Frame(dfmt, dw, dh), dp, flags, interlaced)
    p = dynamic(flags)
    ds = DeviceFrame(sfmt, sw, sh).upload(host_copy(src))
    dd = DeviceFrame(dfmt, dw, dh)
    fl = S.AV_FRAME_FLAG_INTERLACED if interlaced else 0
    ds.props, dd.props = dict(sp, flags=fl), dict(dp, flags=fl)
    dd.buf.fill_(0x5A)
    torch.cuda.synchronize()
    assert p.scale_frame(ds, dd) == 0            # the dynamic path returns 0, not a row count (swscale.c:1479)
    p.sync()
    assert_same(dd.download(), ref, "device frames")
    # the same through host frames (staged by the library)
    hs, hd = host_copy(src), HostFrame(dfmt, dw, dh)
    hs.props, hd.props = ds.props, dd.props
    assert p.scale_frame(hs, hd) == 0
    p.sync()
    assert_same(hd, ref, "host frames")
    # a batch: every pair must carry the properties of the first
    dd2 = DeviceFrame(dfmt, dw, dh)
    dd2.props = dd.props
    assert p.scale_frames([ds, ds], [dd, dd2]) == 2
    p.sync()
    assert_same(dd2.download(), ref, "batch")
    other = DeviceFrame(dfmt, dw + 2, dh)
    other.props = dd.props
    assert p.scale_frames([ds, ds], [dd, other]) == -22
    p.sync()
    p.close()


def test_conversion_is_rebuilt_when_the_properties_or_options_change():
    import torch
    sw, sh, dw, dh = 96, 64, 64, 40
    src = OL.fill_random(OL.Frame("yuv420p", sw, sh), 8)
    ds = DeviceFrame("yuv420p", sw, sh).upload(host_copy(src))
    dd = DeviceFrame("rgb24", dw, dh)
    p = dynamic(LA.SWS_BICUBIC | BX)
    for sp, flags in ((dict(color_range="mpeg"), LA.SWS_BICUBIC | BX), (dict(color_range="jpeg"), LA.SWS_BICUBIC | BX),
                      (dict(color_range="jpeg"), LA.SWS_BILINEAR | BX), (dict(color_range="jpeg", colorspace="bt709"), LA.SWS_BILINEAR | BX),
                      (dict(color_range="mpeg"), LA.SWS_BICUBIC | BX)):
        p.fields().flags = flags
        ds.props = sp
        ref = FP.oracle_convert(src, sp, OL.Frame("rgb24", dw, dh), {}, flags)
        assert p.scale_frame(ds, dd) == 0
        p.sync()
        assert_same(dd.download(), ref, f"{sp} {flags:#x}")
    # the deprecated chroma position fields still win (graph.c:430-444)
    p.fields().src_v_chr_pos = 0
    ref = FP.oracle_convert(src, dict(color_range="mpeg"), OL.Frame("rgb24", dw, dh), {}, LA.SWS_BICUBIC | BX, overrides=dict(src_v_chr_pos=0))
    assert p.scale_frame(ds, dd) == 0
    p.sync()
    assert_same(dd.download(), ref, "src_v_chr_pos override")
    p.close()


@pytest.mark.parametrize("fmt", ["yuv420p", "rgb24", "p010le", "yuva444p"])
def test_noop_is_a_plane_copy(fmt):
    import torch
    w, h = 70, 38
    src = OL.fill_random(OL.Frame(fmt, w, h), 3)
    p = dynamic(LA.SWS_BICUBIC)
    ds = DeviceFrame(fmt, w, h).upload(host_copy(src))
    dd = DeviceFrame(fmt, w, h)
    ds.props = dd.props = dict(color_range="mpeg")
    assert p.L.sws_is_noop(C.byref(dd.view()), C.byref(ds.view()))
    assert p.scale_frame(ds, dd) == 0
    p.sync()
    assert p.L.sws_hip_path_name(p.c) == b"noop:copy"
    assert_same(dd.download(), src, "device -> device")
    hd = HostFrame(fmt, w, h)
    hd.props = ds.props
    assert p.scale_frame(ds, hd) == 0            # complete on return: the destination is host memory
    assert_same(hd, src, "device -> host")
    p.close()


class HipDevice:
    """what av_hwdevice_ctx_create(AV_HWDEVICE_TYPE_HIP) + av_hwframe_ctx_init() leave behind, built by hand (the real thing is
    integration/hwcontext_hip.c, exercised in tests/test_hwcontext_module.py)"""

    def __init__(self, L, device=0):
        self.L, self.keep = L, []
        st = C.c_void_p()
        assert L.sws_hip_stream_create(device, C.byref(st)) == 0 and st.value
        self.hip = S.AVHIPDeviceContext(device, st)
        self.dev = S.SwsHWDeviceContext(None, S.AV_HWDEVICE_TYPE_HIP, C.addressof(self.hip), None, None)

    def frame(self, fmt, w, h, props=None):
        """frames_get_buffer: one allocation, planes at aligned offsets (hwcontext_cuda.c:132-197 analogue)"""
        L = self.L
        fc = S.SwsHWFramesContext()
        fc.device_ctx = C.pointer(self.dev)
        fc.format, fc.sw_format, fc.width, fc.height = S.AV_PIX_FMT_HIP, P[fmt], w, h
        ref = S.SwsBufferRef(None, C.addressof(fc), C.sizeof(fc))
        ls, off, total = (C.c_int * 4)(), (C.c_size_t * 4)(), C.c_size_t()
        assert L.sws_hip_image_layout(P[fmt], w, h, 256, ls, off, C.byref(total)) >= 0
        base = C.c_void_p()
        assert L.sws_hip_mem_alloc(self.hip.device, total.value, C.byref(base)) == 0
        assert L.sws_hip_pointer_device(base) == self.hip.device
        v = S.apply_props(S.SwsFrameView(), props)
        v.width, v.height, v.format = w, h, S.AV_PIX_FMT_HIP
        v.hw_frames_ctx = C.pointer(ref)
        for k in range(len(S.plane_layout(fmt, w, h))):
            v.data[k] = base.value + off[k]
            v.linesize[k] = ls[k]
        self.keep += [fc, ref, (base, v)]
        return v

    def transfer(self, fmt, w, h, hwv, host, to_device):
        """transfer_data_to / transfer_data_from: async plane copies on the device's stream (hwcontext_cuda.c:523-655)"""
        for k, (rb, rows) in enumerate(S.plane_layout(fmt, w, h)):
            a = host.planes[k]
            args = (hwv.data[k], hwv.linesize[k], a.ctypes.data, a.strides[0]) if to_device else (a.ctypes.data, a.strides[0], hwv.data[k], hwv.linesize[k])
            assert self.L.sws_hip_copy_plane(self.hip.device, self.hip.stream, *args, rb, rows) == 0
        if not to_device:
            assert self.L.sws_hip_stream_sync(self.hip.device, self.hip.stream) == 0

    def close(self):
        self.L.sws_hip_stream_sync(self.hip.device, self.hip.stream)
        for item in self.keep:
            if isinstance(item, tuple):
                self.L.sws_hip_mem_free(self.hip.device, item[0])
        self.L.sws_hip_stream_destroy(self.hip.device, self.hip.stream)


@pytest.mark.parametrize("legacy", [False, True], ids=["dynamic", "initialised"])
def test_hip_hardware_frames(legacy):
    """AV_PIX_FMT_HIP frames with a hw_frames_ctx: upload, conversion and download are all queued on the device context's
    stream, nothing is synchronised in between (the reference's CUDA filters rely on the same ordering)."""
    L = S.load_library()
    sw, sh, dw, dh = 352, 288, 200, 120
    flags = LA.SWS_BICUBIC | BX
    sp, dp = dict(color_range="mpeg", colorspace="bt709", chroma_location="left"), {}
    src = OL.fill_random(OL.Frame("nv12", sw, sh), 12)
    if legacy:
        ref = OL.Frame("bgra", dw, dh)
        assert OL.Oracle(sw, sh, "nv12", dw, dh, "bgra", flags).scale(src, ref) == dh
        p = SwsContext(sw, sh, "nv12", dw, dh, "bgra", flags)
    else:
        ref = FP.oracle_convert(src, sp, OL.Frame("bgra", dw, dh), dp, flags)
        p = dynamic(flags)
    hw = HipDevice(L)
    s, d = hw.frame("nv12", sw, sh, sp), hw.frame("bgra", dw, dh, dp)
    hs, out = host_copy(src), HostFrame("bgra", dw, dh)
    for rep in range(3):
        hw.transfer("nv12", sw, sh, s, hs, True)
        r = L.sws_scale_frame(p.c, C.byref(d), C.byref(s))
        assert r == (dh if legacy else 0)
        hw.transfer("bgra", dw, dh, d, out, False)
        assert_same(out, ref, f"hardware frames, round {rep}")
        for a in out.planes:
            a[:] = 0
    # a software frame on one side only is refused (swscale.c:1515-1517)
    dd = DeviceFrame("bgra", dw, dh)
    dv = dd.view()
    assert L.sws_scale_frame(p.c, C.byref(dv), C.byref(s)) == -95
    p.sync()
    p.close()
    hw.close()
