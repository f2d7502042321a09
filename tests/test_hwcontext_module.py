"""integration/hwcontext_hip.c (the HWContextType of AV_HWDEVICE_TYPE_HIP) and the core of integration/vf_scale_hip.c,
compiled against the in-repo declaration shim (integration/shim/) and driven through ctypes.

CPU part: the module builds, fills every vtable slot hwcontext_cuda.c:932-955 fills (minus device_derive / frames_uninit,
which it has no use for), and its constraints list is the converter's format list.
GPU part: the life cycle a libavfilter graph puts a hardware frames pool through -- device create, frames context init,
get_buffer from the pool, upload, scale_hip's per-frame conversion (sws_scale_frame() on AV_PIX_FMT_HIP frames), download --
with the oracle as the judge of the pixels, plus pool recycling and the passthrough decision."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import swscale as S
import frame_props as FP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = LA.PIX_FMT
BX = LA.SWS_BITEXACT


class HWContextType(C.Structure):
    """libavutil/hwcontext_internal.h:29-91"""
    _fields_ = [("type", C.c_int), ("name", C.c_char_p), ("pix_fmts", C.POINTER(C.c_int)), ("device_hwctx_size", C.c_size_t),
                ("device_hwconfig_size", C.c_size_t), ("frames_hwctx_size", C.c_size_t)] + \
               [(n, C.c_void_p) for n in ("device_create", "device_derive", "device_init", "device_uninit", "frames_get_constraints",
                                          "frames_init", "frames_uninit", "frames_get_buffer", "transfer_get_formats", "transfer_data_to",
                                          "transfer_data_from", "map_to", "map_from", "frames_derive_to", "frames_derive_from")]


class Constraints(C.Structure):
    _fields_ = [("valid_hw_formats", C.POINTER(C.c_int)), ("valid_sw_formats", C.POINTER(C.c_int)),
                ("min_width", C.c_int), ("min_height", C.c_int), ("max_width", C.c_int), ("max_height", C.c_int)]


class ScaleHIPCore(C.Structure):
    """integration/vf_scale_hip.c"""
    _fields_ = [("sws", C.c_void_p), ("frames_ctx", C.POINTER(S.SwsBufferRef)), ("w", C.c_int), ("h", C.c_int), ("out_fmt", C.c_int),
                ("out_range", C.c_int), ("out_color_matrix", C.c_int), ("out_chroma_loc", C.c_int), ("passthrough", C.c_int)]


@pytest.fixture(scope="module")
def mod(hiplib):
    path = os.path.join(ROOT, "integration", "_build", "libavhip_test.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "integration")])
    M = C.CDLL(path, mode=C.RTLD_GLOBAL)
    ref = C.POINTER(S.SwsBufferRef)
    fr = C.POINTER(S.SwsFrameView)
    M.shim_hwdevice_ctx_create.argtypes = [C.POINTER(ref), C.c_void_p, C.c_char_p]
    M.av_hwframe_ctx_alloc.argtypes = [ref]; M.av_hwframe_ctx_alloc.restype = ref
    M.av_hwframe_ctx_init.argtypes = [ref]
    M.av_hwframe_get_buffer.argtypes = [ref, fr, C.c_int]
    M.av_hwframe_transfer_data.argtypes = [fr, fr, C.c_int]
    M.av_frame_alloc.restype = fr
    M.av_frame_unref.argtypes = [fr]; M.av_frame_unref.restype = None
    M.av_frame_free.argtypes = [C.POINTER(fr)]; M.av_frame_free.restype = None
    M.av_buffer_unref.argtypes = [C.POINTER(ref)]; M.av_buffer_unref.restype = None
    M.av_free.argtypes = [C.c_void_p]; M.av_free.restype = None
    M.scale_hip_core_init.argtypes = [C.POINTER(ScaleHIPCore), ref, C.c_int, C.c_int, C.c_int, C.c_uint]
    M.scale_hip_core_frame.argtypes = [C.POINTER(ScaleHIPCore), fr, fr]
    M.scale_hip_core_uninit.argtypes = [C.POINTER(ScaleHIPCore)]; M.scale_hip_core_uninit.restype = None
    return M


def vtable(M):
    return HWContextType.in_dll(M, "ff_hwcontext_type_hip")


def test_vtable_is_filled_like_the_cuda_one(mod):
    t = vtable(mod)
    assert t.type == S.AV_HWDEVICE_TYPE_HIP and t.name == b"HIP"
    assert [t.pix_fmts[0], t.pix_fmts[1]] == [S.AV_PIX_FMT_HIP, -1]
    assert t.device_hwctx_size >= C.sizeof(S.AVHIPDeviceContext) and t.frames_hwctx_size > 0
    for slot in ("device_create", "device_init", "device_uninit", "frames_get_constraints", "frames_init", "frames_get_buffer",
                 "transfer_get_formats", "transfer_data_to", "transfer_data_from"):
        assert getattr(t, slot), slot
    assert not t.map_to and not t.map_from and not t.frames_derive_to


def test_constraints_list_the_converters_formats(mod, hiplib):
    t = vtable(mod)
    fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Constraints))(t.frames_get_constraints)
    c = Constraints()
    assert fn(None, None, C.byref(c)) == 0
    assert [c.valid_hw_formats[0], c.valid_hw_formats[1]] == [S.AV_PIX_FMT_HIP, -1]
    got = []
    while c.valid_sw_formats[len(got)] != -1:
        got.append(c.valid_sw_formats[len(got)])
    want = [f for f in range(S.AV_PIX_FMT_HIP) if hiplib.sws_isSupportedInput(f) and hiplib.sws_isSupportedOutput(f)]
    assert got == want and P["yuv420p"] in got and P["p010le"] in got and P["bgra"] in got and len(got) > 100
    mod.av_free(c.valid_hw_formats)
    mod.av_free(c.valid_sw_formats)


def _host_frame(fr):
    """an AVFrame over the planes of an oracle_lib.Frame / HostFrame"""
    v = S.apply_props(S.SwsFrameView(), getattr(fr, "props", None))
    for i, a in enumerate(fr.planes):
        v.data[i] = a.ctypes.data
        v.linesize[i] = a.strides[0]
    v.width, v.height, v.format = fr.w, fr.h, P[fr.fmt]
    return v


def _frames_ctx(M, dev, fmt, w, h):
    ref = M.av_hwframe_ctx_alloc(dev)
    assert ref
    fc = C.cast(ref.contents.data, C.POINTER(S.SwsHWFramesContext)).contents
    fc.format, fc.sw_format, fc.width, fc.height = S.AV_PIX_FMT_HIP, P[fmt], w, h
    return ref


@pytest.mark.gpu
def test_pool_upload_scale_download(mod):
    M = mod
    L = S.load_library()
    dev = C.POINTER(S.SwsBufferRef)()
    assert M.shim_hwdevice_ctx_create(C.byref(dev), C.addressof(vtable(M)), b"99") == -19       # ENODEV: no such device
    assert M.shim_hwdevice_ctx_create(C.byref(dev), C.addressof(vtable(M)), b"0") == 0
    hip = C.cast(C.cast(dev.contents.data, C.POINTER(S.SwsHWDeviceContext)).contents.hwctx, C.POINTER(S.AVHIPDeviceContext)).contents
    assert hip.device == 0 and hip.stream
    bad = _frames_ctx(M, dev, "yuv420p", 64, 64)
    C.cast(bad.contents.data, C.POINTER(S.SwsHWFramesContext)).contents.sw_format = 11           # pal8: not a converter format
    assert M.av_hwframe_ctx_init(bad) == -38                                                     # ENOSYS
    M.av_buffer_unref(C.byref(bad))

    sw, sh, dw, dh = 352, 288, 200, 120
    flags = LA.SWS_BICUBIC | BX
    sp = dict(color_range="mpeg", colorspace="bt709", chroma_location="left")
    src = OL.fill_random(OL.Frame("yuv420p", sw, sh), 21)
    src.props = sp
    in_ref = _frames_ctx(M, dev, "yuv420p", sw, sh)
    assert M.av_hwframe_ctx_init(in_ref) == 0
    core = ScaleHIPCore(out_range=-1, out_color_matrix=-1, out_chroma_loc=-1, passthrough=1)
    assert M.scale_hip_core_init(C.byref(core), in_ref, dw, dh, P["bgra"], flags) == 0
    ref = FP.oracle_convert(src, sp, OL.Frame("bgra", dw, dh), {}, flags)

    seen = set()
    for rep in range(4):
        fin, fout = M.av_frame_alloc(), M.av_frame_alloc()
        assert M.av_hwframe_get_buffer(in_ref, fin, 0) == 0
        f = fin.contents
        assert f.format == S.AV_PIX_FMT_HIP and (f.width, f.height) == (sw, sh) and f.linesize[0] % 256 == 0 and f.data[0] % 256 == 0
        assert L.sws_hip_pointer_device(f.data[0]) == 0
        seen.add(f.data[0])
        hv = _host_frame(src)
        assert M.av_hwframe_transfer_data(fin, C.byref(hv), 0) == 0          # upload: queued, not waited for
        S.apply_props(f, sp)
        assert M.scale_hip_core_frame(C.byref(core), fout, fin) == 0         # conversion: queued on the same stream
        o = fout.contents
        assert (o.width, o.height, o.format, o.color_range) == (dw, dh, S.AV_PIX_FMT_HIP, 1)
        out = S.HostFrame("bgra", dw, dh)
        ov = _host_frame(out)
        assert M.av_hwframe_transfer_data(C.byref(ov), fout, 0) == 0         # download: complete on return
        for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
            assert np.array_equal(a[:, :out.row_bytes[i]], b[:, :out.row_bytes[i]]), (rep, i)
        M.av_frame_free(C.byref(fin))
        M.av_frame_free(C.byref(fout))
    assert len(seen) == 1                                                    # the pool hands the released buffer out again

    # passthrough: same size, same format, same properties -> the input frame is handed on
    same = ScaleHIPCore(out_range=-1, out_color_matrix=-1, out_chroma_loc=-1, passthrough=1)
    assert M.scale_hip_core_init(C.byref(same), in_ref, sw, sh, -1, flags) == 0
    fin, fout = M.av_frame_alloc(), M.av_frame_alloc()
    assert M.av_hwframe_get_buffer(in_ref, fin, 0) == 0
    S.apply_props(fin.contents, sp)
    assert M.scale_hip_core_frame(C.byref(same), fout, fin) == 1 and not fout.contents.data[0]
    # ... unless an output property differs: mpeg -> jpeg range is a real conversion
    same.out_range = 2
    assert M.scale_hip_core_frame(C.byref(same), fout, fin) == 0 and fout.contents.color_range == 2
    M.av_frame_free(C.byref(fin))
    M.av_frame_free(C.byref(fout))

    M.scale_hip_core_uninit(C.byref(core))
    M.scale_hip_core_uninit(C.byref(same))
    M.av_buffer_unref(C.byref(in_ref))
    M.av_buffer_unref(C.byref(dev))
    assert M.shim_live_buffers() == 0                                        # every hipMalloc()ed block and context went back
