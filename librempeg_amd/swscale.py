"""ctypes binding of libswscale_hip.so (include/swscale_hip.h).

Mirrors the reference's public API names and argument meaning (libswscale/swscale.h):
sws_getContext / sws_setColorspaceDetails / sws_scale / sws_freeContext, plus the new
sws_scale_frames() batch entry point.  Fails loudly if the HIP library is missing: there is
no CPU fallback in the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SWS_FAST_BILINEAR, SWS_BILINEAR, SWS_BICUBIC, SWS_X, SWS_POINT, SWS_AREA = 1, 2, 4, 8, 16, 32
SWS_BICUBLIN, SWS_GAUSS, SWS_SINC, SWS_LANCZOS, SWS_SPLINE = 64, 128, 256, 512, 1024
SWS_PRINT_INFO, SWS_FULL_CHR_H_INT, SWS_FULL_CHR_H_INP = 1 << 12, 1 << 13, 1 << 14
SWS_ACCURATE_RND, SWS_BITEXACT = 1 << 18, 1 << 19
SWS_CS_ITU709, SWS_CS_ITU601, SWS_CS_DEFAULT, SWS_CS_BT2020 = 1, 5, 5, 9

# enum AVPixelFormat values (libavutil/pixfmt.h)
# name: (AVPixelFormat value, layout, log2 chroma w, log2 chroma h, bytes per sample)
# layout: "planar" (3 planes), "semi" (Y + interleaved UV), "packed" (bytes per sample = bytes per pixel),
#         "rgbp" (3 full-size planes), "gray" (1 plane)
_FORMATS = {
    "yuv420p": (0, "planar", 1, 1, 1), "yuvj420p": (12, "planar", 1, 1, 1), "yuv422p": (4, "planar", 1, 0, 1),
    "yuvj422p": (13, "planar", 1, 0, 1), "yuv444p": (5, "planar", 0, 0, 1), "yuvj444p": (14, "planar", 0, 0, 1),
    "yuva420p": (33, "planara", 1, 1, 1), "yuva422p": (78, "planara", 1, 0, 1), "yuva444p": (79, "planara", 0, 0, 1),
    "yuv410p": (6, "planar", 2, 2, 1), "yuv411p": (7, "planar", 2, 0, 1), "yuv440p": (31, "planar", 0, 1, 1),
    "yuva420p9le": (81, "planara", 1, 1, 2), "yuva420p10le": (87, "planara", 1, 1, 2), "yuva420p16le": (93, "planara", 1, 1, 2), "yuva422p9le": (83, "planara", 1, 0, 2), "yuva422p10le": (89, "planara", 1, 0, 2), "yuva422p12le": (185, "planara", 1, 0, 2), "yuva422p16le": (95, "planara", 1, 0, 2), "yuva444p9le": (85, "planara", 0, 0, 2), "yuva444p10le": (91, "planara", 0, 0, 2), "yuva444p12le": (187, "planara", 0, 0, 2), "yuva444p16le": (97, "planara", 0, 0, 2),
    "grayf32le": (183, "gray", 0, 0, 4), "ya8": (56, "packed", 0, 0, 2), "ya16le": (110, "packed", 0, 0, 4),
    "yuvj440p": (32, "planar", 0, 1, 1), "monow": (9, "mono", 0, 0, 1), "monob": (10, "mono", 0, 0, 1),
    # inputs only: float / half-float pictures and the packed 4:1:1 layout
    "rgbf32le": (218, "packed", 0, 0, 12), "rgbf32be": (217, "packed", 0, 0, 12), "rgbf16le": (234, "packed", 0, 0, 6), "rgbf16be": (233, "packed", 0, 0, 6),
    "rgbaf16le": (207, "packed", 0, 0, 8), "rgbaf16be": (206, "packed", 0, 0, 8), "grayf16le": (248, "gray", 0, 0, 2), "grayf16be": (247, "gray", 0, 0, 2),
    "yaf32le": (253, "packed", 0, 0, 8), "yaf32be": (252, "packed", 0, 0, 8), "yaf16le": (255, "packed", 0, 0, 4), "yaf16be": (254, "packed", 0, 0, 4),
    "gbrpf16le": (244, "rgbp", 0, 0, 2), "gbrpf16be": (243, "rgbp", 0, 0, 2), "gbrapf16le": (246, "rgbap", 0, 0, 2), "gbrapf16be": (245, "rgbap", 0, 0, 2),
    "uyyvyy411": (16, "packed411", 2, 0, 1), "pal8": (11, "pal", 0, 0, 1),
    "bayer_bggr8": (139, "gray", 0, 0, 1), "bayer_rggb8": (140, "gray", 0, 0, 1), "bayer_gbrg8": (141, "gray", 0, 0, 1), "bayer_grbg8": (142, "gray", 0, 0, 1),
    "bayer_bggr16le": (143, "gray", 0, 0, 2), "bayer_bggr16be": (144, "gray", 0, 0, 2), "bayer_rggb16le": (145, "gray", 0, 0, 2), "bayer_rggb16be": (146, "gray", 0, 0, 2),
    "bayer_gbrg16le": (147, "gray", 0, 0, 2), "bayer_gbrg16be": (148, "gray", 0, 0, 2), "bayer_grbg16le": (149, "gray", 0, 0, 2), "bayer_grbg16be": (150, "gray", 0, 0, 2),
    "bgr8": (17, "packed", 0, 0, 1), "bgr4": (18, "nibble", 0, 0, 1), "bgr4_byte": (19, "packed", 0, 0, 1), "rgb8": (20, "packed", 0, 0, 1), "rgb4": (21, "nibble", 0, 0, 1), "rgb4_byte": (22, "packed", 0, 0, 1),
    "xyz12le": (99, "packed", 0, 0, 6), "yuvj411p": (138, "planar", 2, 0, 1), "nv20le": (102, "semi", 1, 0, 2),
    "gbrp10msble": (263, "rgbp", 0, 0, 2), "gbrp12msble": (265, "rgbp", 0, 0, 2),
    "yuv420p9le": (60, "planar", 1, 1, 2), "yuv422p9le": (70, "planar", 1, 0, 2), "yuv444p9le": (66, "planar", 0, 0, 2),
    "yuv420p10le": (62, "planar", 1, 1, 2), "yuv422p10le": (64, "planar", 1, 0, 2), "yuv444p10le": (68, "planar", 0, 0, 2),
    "yuv440p10le": (151, "planar", 0, 1, 2),
    "yuv420p12le": (123, "planar", 1, 1, 2), "yuv422p12le": (127, "planar", 1, 0, 2), "yuv444p12le": (131, "planar", 0, 0, 2),
    "yuv440p12le": (153, "planar", 0, 1, 2),
    "yuv420p14le": (125, "planar", 1, 1, 2), "yuv422p14le": (129, "planar", 1, 0, 2), "yuv444p14le": (133, "planar", 0, 0, 2),
    "yuv420p16le": (45, "planar", 1, 1, 2), "yuv422p16le": (47, "planar", 1, 0, 2), "yuv444p16le": (49, "planar", 0, 0, 2),
    "nv12": (23, "semi", 1, 1, 1), "nv21": (24, "semi", 1, 1, 1), "nv16": (101, "semi", 1, 0, 1),
    "nv24": (188, "semi", 0, 0, 1), "nv42": (189, "semi", 0, 0, 1),
    "p010le": (158, "semi", 1, 1, 2), "p012le": (209, "semi", 1, 1, 2), "p016le": (169, "semi", 1, 1, 2),
    "p210le": (198, "semi", 1, 0, 2), "p212le": (222, "semi", 1, 0, 2), "p216le": (202, "semi", 1, 0, 2),
    "p410le": (200, "semi", 0, 0, 2), "p412le": (224, "semi", 0, 0, 2), "p416le": (204, "semi", 0, 0, 2),
    "yuyv422": (1, "packed422", 1, 0, 1), "uyvy422": (15, "packed422", 1, 0, 1), "yvyu422": (108, "packed422", 1, 0, 1),
    "rgb48le": (35, "packed", 0, 0, 6), "bgr48le": (58, "packed", 0, 0, 6), "rgba64le": (105, "packed", 0, 0, 8), "bgra64le": (107, "packed", 0, 0, 8),
    "rgb565le": (37, "packed", 0, 0, 2), "rgb555le": (39, "packed", 0, 0, 2), "rgb444le": (52, "packed", 0, 0, 2),
    "bgr565le": (41, "packed", 0, 0, 2), "bgr555le": (43, "packed", 0, 0, 2), "bgr444le": (54, "packed", 0, 0, 2),
    "yuv444p10msble": (259, "planar", 0, 0, 2), "yuv444p12msble": (261, "planar", 0, 0, 2),
    "vyu444": (230, "packed", 0, 0, 3), "uyva": (229, "packed", 0, 0, 4), "ayuv": (228, "packed", 0, 0, 4), "vuya": (205, "packed", 0, 0, 4), "vuyx": (208, "packed", 0, 0, 4),
    "y210le": (192, "packed422", 1, 0, 2), "y212le": (212, "packed422", 1, 0, 2), "y216le": (240, "packed422", 1, 0, 2),
    "x2rgb10le": (193, "packed", 0, 0, 4), "x2bgr10le": (195, "packed", 0, 0, 4), "xv30le": (214, "packed", 0, 0, 4), "v30xle": (232, "packed", 0, 0, 4), "xv36le": (216, "packed", 0, 0, 8), "xv48le": (242, "packed", 0, 0, 8), "ayuv64le": (155, "packed", 0, 0, 8),
    "rgb24": (2, "packed", 0, 0, 3), "bgr24": (3, "packed", 0, 0, 3),
    "argb": (25, "packed", 0, 0, 4), "rgba": (26, "packed", 0, 0, 4), "abgr": (27, "packed", 0, 0, 4), "bgra": (28, "packed", 0, 0, 4),
    "0rgb": (118, "packed", 0, 0, 4), "rgb0": (119, "packed", 0, 0, 4), "0bgr": (120, "packed", 0, 0, 4), "bgr0": (121, "packed", 0, 0, 4),
    "gbrp": (71, "rgbp", 0, 0, 1), "gbrpf32le": (175, "rgbp", 0, 0, 4),
    "gbrap": (111, "rgbap", 0, 0, 1), "gbrap10le": (163, "rgbap", 0, 0, 2), "gbrap12le": (161, "rgbap", 0, 0, 2), "gbrap14le": (226, "rgbap", 0, 0, 2),
    "gbrap16le": (113, "rgbap", 0, 0, 2), "gbrapf32le": (177, "rgbap", 0, 0, 4),
    "gbrp9le": (73, "rgbp", 0, 0, 2), "gbrp10le": (75, "rgbp", 0, 0, 2), "gbrp12le": (135, "rgbp", 0, 0, 2),
    "gbrp14le": (137, "rgbp", 0, 0, 2), "gbrp16le": (77, "rgbp", 0, 0, 2),
    "gray8": (8, "gray", 0, 0, 1), "gray9le": (173, "gray", 0, 0, 2), "gray10le": (168, "gray", 0, 0, 2),
    "gray12le": (166, "gray", 0, 0, 2), "gray14le": (181, "gray", 0, 0, 2), "gray16le": (30, "gray", 0, 0, 2),
}

# big-endian twins: same layout as the little-endian format, AVPixelFormat value from libavutil/pixfmt.h
_BE_VALUES = {"gbrap10be": 162, "gbrap12be": 160, "gbrap14be": 225, "gbrap16be": 112, "gbrapf32be": 176, "ya16be": 109, "grayf32be": 182, "yuva420p9be": 80, "yuva420p10be": 86, "yuva420p16be": 92, "yuva422p9be": 82, "yuva422p10be": 88, "yuva422p12be": 184, "yuva422p16be": 94, "yuva444p9be": 84, "yuva444p10be": 90, "yuva444p12be": 186, "yuva444p16be": 96, "xyz12be": 100, "nv20be": 103, "gbrp10msbbe": 262, "gbrp12msbbe": 264, "xv36be": 215, "xv48be": 241, "ayuv64be": 156, "yuv444p10msbbe": 258, "yuv444p12msbbe": 260, "rgb565be": 36, "rgb555be": 38, "rgb444be": 53, "bgr565be": 40, "bgr555be": 42, "bgr444be": 55, "yuv420p9be": 59, "yuv420p10be": 61, "yuv420p12be": 122, "yuv420p14be": 124, "yuv420p16be": 46, "yuv422p9be": 69, "yuv422p10be": 63, "yuv422p12be": 126, "yuv422p14be": 128, "yuv422p16be": 48, "yuv444p9be": 65, "yuv444p10be": 67, "yuv444p12be": 130, "yuv444p14be": 132, "yuv444p16be": 50, "yuv440p10be": 152, "yuv440p12be": 154, "gray9be": 172, "gray10be": 167, "gray12be": 165, "gray14be": 180, "gray16be": 29, "gbrp9be": 72, "gbrp10be": 74, "gbrp12be": 134, "gbrp14be": 136, "gbrp16be": 76, "gbrpf32be": 174, "p010be": 159, "p012be": 210, "p016be": 170, "p210be": 197, "p212be": 221, "p216be": 201, "p410be": 199, "p412be": 223, "p416be": 203, "rgb48be": 34, "bgr48be": 57, "rgba64be": 104, "bgra64be": 106}
for _n, _v in list(_BE_VALUES.items()):
    _le = _FORMATS[_n[:-2] + "le"]
    _FORMATS[_n] = (_v,) + _le[1:]


def plane_layout(fmt, w, h):
    """[(visible_bytes_per_row, rows)] per plane."""
    _, kind, lw, lh, bps = _FORMATS[fmt]
    cw, ch = -(-w >> lw), -(-h >> lh)
    if kind == "planar":
        return [(bps * w, h), (bps * cw, ch), (bps * cw, ch)]
    if kind == "planara":
        return [(bps * w, h), (bps * cw, ch), (bps * cw, ch), (bps * w, h)]
    if kind == "semi":
        return [(bps * w, h), (2 * bps * cw, ch)]
    if kind == "packed422":      # Y0 U Y1 V groups: 4 bytes per pixel pair (libavutil/imgutils.c av_image_get_linesize)
        return [(4 * bps * cw, h)]
    if kind == "mono":           # 1 bit per pixel, MSB first
        return [((w + 7) >> 3, h)]
    if kind == "packed411":      # U Y Y V Y Y groups: 6 bytes per 4 pixels (av_image_get_linesize: step 6 over the chroma-shifted width)
        return [(6 * cw, h)]
    if kind == "pal":            # index plane + 256 native-endian 0xAARRGGBB words in data[1]
        return [(w, h), (1024, 1)]
    if kind == "nibble":         # rgb4 / bgr4: 4 bits per pixel, two pixels per byte
        return [((4 * w + 7) >> 3, h)]
    if kind == "rgbp":
        return [(bps * w, h)] * 3
    if kind == "rgbap":
        return [(bps * w, h)] * 4
    return [(bps * w, h)]   # packed, gray
PIX_FMT = {k: v[0] for k, v in _FORMATS.items()}
_FMT_NAME = {v: k for k, v in PIX_FMT.items()}


def library_path():
    # SWS_HIP_LIBRARY: load another build of the same library (the -DSWS_HIP_PROFILING build of tools/build_profiling.sh)
    return os.environ.get("SWS_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libswscale_hip.so")


def build_library(force=False):
    """Compile libswscale_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", src, "clean"])
    # one object per translation unit; "prefixed" = libswship.so, the same code exported as swship_* (include/swscale_hip_prefix.h)
    subprocess.check_call(["make", "-s", "-j", str(max(2, os.cpu_count() or 2)), "-C", src, "all", "prefixed"])
    return library_path()


class SwsBufferRef(C.Structure):
    """libavutil/buffer.h:82-95 AVBufferRef"""
    _fields_ = [("buffer", C.c_void_p), ("data", C.c_void_p), ("size", C.c_size_t)]


class SwsRational(C.Structure):
    _fields_ = [("num", C.c_int), ("den", C.c_int)]


class SwsFrameSideData(C.Structure):
    """libavutil/frame.h:236-251 AVFrameSideData"""
    _fields_ = [("type", C.c_int), ("data", C.c_void_p), ("size", C.c_size_t), ("metadata", C.c_void_p), ("buf", C.c_void_p)]


class SwsChannelLayout(C.Structure):
    _fields_ = [("order", C.c_int), ("nb_channels", C.c_int), ("mask", C.c_uint64), ("opaque", C.c_void_p)]


class SwsFrameView(C.Structure):
    """Field-for-field mirror of AVFrame (libavutil/frame.h:472-828), see include/swscale_hip.h."""
    _fields_ = [("data", C.c_void_p * 8), ("linesize", C.c_int * 8), ("extended_data", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int), ("nb_samples", C.c_int), ("format", C.c_int),
                ("pict_type", C.c_int), ("sample_aspect_ratio", SwsRational), ("pts", C.c_int64), ("pkt_dts", C.c_int64),
                ("time_base", SwsRational), ("quality", C.c_int), ("opaque", C.c_void_p), ("repeat_pict", C.c_int),
                ("sample_rate", C.c_int), ("buf", C.c_void_p * 8), ("extended_buf", C.c_void_p), ("nb_extended_buf", C.c_int),
                ("side_data", C.POINTER(C.POINTER(SwsFrameSideData))), ("nb_side_data", C.c_int), ("flags", C.c_int),
                ("color_range", C.c_int), ("color_primaries", C.c_int), ("color_trc", C.c_int), ("colorspace", C.c_int),
                ("chroma_location", C.c_int), ("best_effort_timestamp", C.c_int64), ("metadata", C.c_void_p),
                ("decode_error_flags", C.c_int), ("hw_frames_ctx", C.POINTER(SwsBufferRef)), ("opaque_ref", C.c_void_p),
                ("crop_top", C.c_size_t), ("crop_bottom", C.c_size_t), ("crop_left", C.c_size_t), ("crop_right", C.c_size_t),
                ("private_ref", C.c_void_p), ("ch_layout", SwsChannelLayout), ("duration", C.c_int64), ("alpha_mode", C.c_int)]


class AVHIPDeviceContext(C.Structure):
    """include/hwcontext_hip.h"""
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p)]


class SwsHWDeviceContext(C.Structure):
    """libavutil/hwcontext.h:63-106 AVHWDeviceContext"""
    _fields_ = [("av_class", C.c_void_p), ("type", C.c_int), ("hwctx", C.c_void_p), ("free", C.c_void_p), ("user_opaque", C.c_void_p)]


class SwsHWFramesContext(C.Structure):
    """libavutil/hwcontext.h:118-221 AVHWFramesContext"""
    _fields_ = [("av_class", C.c_void_p), ("device_ref", C.c_void_p), ("device_ctx", C.POINTER(SwsHWDeviceContext)), ("hwctx", C.c_void_p),
                ("free", C.c_void_p), ("user_opaque", C.c_void_p), ("pool", C.c_void_p), ("initial_pool_size", C.c_int),
                ("format", C.c_int), ("sw_format", C.c_int), ("width", C.c_int), ("height", C.c_int)]


AV_PIX_FMT_HIP = 268
AV_HWDEVICE_TYPE_HIP = 15
AV_FRAME_FLAG_INTERLACED = 1 << 3
COL_RANGE = {"unspecified": 0, "mpeg": 1, "tv": 1, "jpeg": 2, "pc": 2}
CHROMA_LOC = {"unspecified": 0, "left": 1, "center": 2, "topleft": 3, "top": 4, "bottomleft": 5, "bottom": 6}
COL_SPC = {"rgb": 0, "bt709": 1, "unspecified": 2, "fcc": 4, "bt470bg": 5, "smpte170m": 6, "smpte240m": 7, "bt2020nc": 9}


def apply_props(v, props):
    """frame properties (AVFrame fields by name: color_range=, colorspace=, chroma_location=, color_primaries=, color_trc=, flags=)"""
    v.color_primaries = v.color_trc = v.colorspace = 2      # *_UNSPECIFIED, as av_frame_alloc() leaves them (frame.c get_frame_defaults)
    for k, val in (props or {}).items():
        if isinstance(val, str):
            val = {"color_range": COL_RANGE, "chroma_location": CHROMA_LOC, "colorspace": COL_SPC}[k][val]
        setattr(v, k, val)
    return v


class SwsVector(C.Structure):
    """swscale.h:478-481"""
    _fields_ = [("coeff", C.POINTER(C.c_double)), ("length", C.c_int)]


class SwsFilter(C.Structure):
    """swscale.h:484-489"""
    _fields_ = [("lumH", C.POINTER(SwsVector)), ("lumV", C.POINTER(SwsVector)),
                ("chrH", C.POINTER(SwsVector)), ("chrV", C.POINTER(SwsVector))]


def vec_to_list(v):
    return [v.contents.coeff[i] for i in range(v.contents.length)]


def filter_to_dict(f):
    """SwsFilter* -> {"lumH": [...], ...} (the shape tests/oracle_lib.Oracle(src_filter=) takes)"""
    return {n: vec_to_list(getattr(f.contents, n)) for n in ("lumH", "lumV", "chrH", "chrV")}


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own ROCm runtime (libamdhip64.so.7 / libhsa-runtime64); load it FIRST so that
    # libswscale_hip.so binds to the same HIP/HSA instance -- two HSA runtimes in one process lose the GPU.
    # (SWS_HIP_NO_TORCH=1: a process that will not use torch -- the library then binds to the system ROCm runtime of its RUNPATH)
    if os.environ.get("SWS_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with librempeg_amd.build_library() "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    L.sws_getContext.restype = vp
    L.sws_getContext.argtypes = [ci, ci, ci, ci, ci, ci, ci, vp, vp, C.POINTER(C.c_double)]
    L.sws_alloc_context.restype = vp
    L.sws_init_context.argtypes = [vp, vp, vp]
    L.sws_freeContext.argtypes = [vp]
    L.sws_getCoefficients.restype = C.POINTER(ci)
    L.sws_getCoefficients.argtypes = [ci]
    L.sws_setColorspaceDetails.argtypes = [vp, C.POINTER(ci), ci, C.POINTER(ci), ci, ci, ci, ci]
    L.sws_scale.argtypes = [vp, C.POINTER(vp), C.POINTER(ci), ci, ci, C.POINTER(vp), C.POINTER(ci)]
    L.sws_scale_frame.argtypes = [vp, C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_scale_frames.argtypes = [vp, C.POINTER(C.POINTER(SwsFrameView)), C.POINTER(C.POINTER(SwsFrameView)), ci]
    L.sws_isSupportedInput.argtypes = [ci]
    L.sws_isSupportedOutput.argtypes = [ci]
    L.sws_hip_device_count.restype = ci
    L.sws_hip_set_device.argtypes = [vp, ci]
    L.sws_hip_set_stream.argtypes = [vp, vp]
    L.sws_hip_get_stream.restype = vp
    L.sws_hip_get_stream.argtypes = [vp]
    L.sws_hip_sync.argtypes = [vp]
    # device-level helpers of include/hwcontext_hip.h
    L.sws_hip_mem_alloc.argtypes = [ci, C.c_size_t, C.POINTER(vp)]
    L.sws_hip_mem_free.argtypes = [ci, vp]
    L.sws_hip_mem_free.restype = None
    L.sws_hip_stream_create.argtypes = [ci, C.POINTER(vp)]
    L.sws_hip_stream_destroy.argtypes = [ci, vp]
    L.sws_hip_stream_destroy.restype = None
    L.sws_hip_stream_sync.argtypes = [ci, vp]
    L.sws_hip_copy_plane.argtypes = [ci, vp, vp, ci, vp, ci, ci, ci]
    L.sws_hip_pointer_device.argtypes = [vp]
    L.sws_hip_frames_format_supported.argtypes = [ci]
    L.sws_hip_frame_alloc.argtypes = [C.POINTER(SwsFrameView), ci, ci, ci, ci]
    L.sws_hip_frame_free.argtypes = [C.POINTER(SwsFrameView)]
    L.sws_hip_frame_upload.argtypes = [vp, C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_hip_frame_download.argtypes = [vp, C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_hip_image_layout.argtypes = [ci, ci, ci, ci, C.POINTER(ci), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.sws_hip_tables_size.restype = C.c_size_t
    L.sws_hip_tables_size.argtypes = [vp]
    L.sws_hip_tables_export.argtypes = [vp, vp, C.c_size_t]
    L.sws_hip_tables_import.argtypes = [vp, vp, C.c_size_t]
    L.sws_hip_path_name.restype = C.c_char_p
    L.sws_hip_path_name.argtypes = [vp]
    L.sws_hip_kernel_name.restype = C.c_char_p
    L.sws_hip_kernel_name.argtypes = [vp]
    L.sws_hip_debug_check.argtypes = [vp, C.c_char_p, ci]
    L.sws_hip_get_filter.argtypes = [vp, ci, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(ci)]
    L.sws_hip_get_tables.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(ci), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
    L.sws_hip_last_kernel_ms.restype = C.c_double
    L.sws_hip_last_kernel_ms.argtypes = [vp]
    L.sws_hip_set_timing.argtypes = [vp, ci]
    L.sws_hip_set_option.argtypes = [vp, C.c_char_p, ci]
    L.sws_hip_get_device.argtypes = [vp]
    L.sws_hip_plan_shards.argtypes = [ci, C.POINTER(ci), C.POINTER(ci), ci, ci, C.POINTER(ci)]
    L.swscale_version.restype = C.c_uint
    cd, cf, cu = C.c_double, C.c_float, C.c_uint
    L.sws_allocVec.restype = C.POINTER(SwsVector)
    L.sws_allocVec.argtypes = [ci]
    L.sws_getGaussianVec.restype = C.POINTER(SwsVector)
    L.sws_getGaussianVec.argtypes = [cd, cd]
    L.sws_scaleVec.argtypes = [C.POINTER(SwsVector), cd]
    L.sws_scaleVec.restype = None
    L.sws_normalizeVec.argtypes = [C.POINTER(SwsVector), cd]
    L.sws_normalizeVec.restype = None
    L.sws_freeVec.argtypes = [C.POINTER(SwsVector)]
    L.sws_freeVec.restype = None
    L.sws_getDefaultFilter.restype = C.POINTER(SwsFilter)
    L.sws_getDefaultFilter.argtypes = [cf, cf, cf, cf, cf, cf, ci]
    L.sws_freeFilter.argtypes = [C.POINTER(SwsFilter)]
    L.sws_freeFilter.restype = None
    L.sws_convertPalette8ToPacked32.argtypes = [vp, vp, ci, vp]
    L.sws_convertPalette8ToPacked32.restype = None
    L.sws_convertPalette8ToPacked24.argtypes = [vp, vp, ci, vp]
    L.sws_convertPalette8ToPacked24.restype = None
    L.sws_get_class.restype = vp
    for n in ("sws_test_format", "sws_test_colorspace", "sws_test_primaries", "sws_test_transfer"):
        getattr(L, n).argtypes = [ci, ci]
    L.sws_test_hw_format.argtypes = [ci]
    L.sws_test_frame.argtypes = [C.POINTER(SwsFrameView), ci]
    L.sws_is_noop.argtypes = [C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_frame_setup.argtypes = [vp, C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_frame_start.argtypes = [vp, C.POINTER(SwsFrameView), C.POINTER(SwsFrameView)]
    L.sws_frame_end.argtypes = [vp]
    L.sws_frame_end.restype = None
    L.sws_send_slice.argtypes = [vp, cu, cu]
    L.sws_receive_slice.argtypes = [vp, cu, cu]
    L.sws_receive_slice_alignment.argtypes = [vp]
    L.sws_receive_slice_alignment.restype = cu
    _LIB = L
    return L


def image_layout(fmt, w, h, align=256):
    """(linesizes, offsets, total_bytes) of the library's linear frame layout."""
    L = load_library()
    ls = (C.c_int * 4)()
    off = (C.c_size_t * 4)()
    tot = C.c_size_t()
    r = L.sws_hip_image_layout(PIX_FMT[fmt], w, h, align, ls, off, C.byref(tot))
    if r < 0:
        raise ValueError(fmt)
    return list(ls), list(off), tot.value


class HostFrame:
    """Host image: per-plane 2-D uint8 numpy arrays (row stride = array stride)."""

    def __init__(self, fmt, w, h, align=64):
        self.fmt, self.w, self.h = fmt, w, h
        self.row_bytes = [rb for rb, _ in plane_layout(fmt, w, h)]
        self.planes = []
        for rb, rows in plane_layout(fmt, w, h):
            stride = (rb + align - 1) // align * align if align else rb
            self.planes.append(np.zeros((rows, stride), dtype=np.uint8))

    def ptrs(self):
        p = (C.c_void_p * 4)()
        s = (C.c_int * 4)()
        for i, a in enumerate(self.planes):
            p[i] = a.ctypes.data
            s[i] = a.strides[0]
        return p, s

    def view(self):
        v = SwsFrameView()
        for i, a in enumerate(self.planes):
            v.data[i] = a.ctypes.data
            v.linesize[i] = a.strides[0]
        v.width, v.height, v.format = self.w, self.h, PIX_FMT[self.fmt]
        return apply_props(v, getattr(self, "props", None))

    def visible(self):
        return b"".join(a[:, :rb].tobytes() for a, rb in zip(self.planes, self.row_bytes))


class DeviceFrame:
    """HBM-resident frame backed by ONE torch uint8 tensor (planes at 256-byte aligned offsets).

    The tensor is allocated with GUARD bytes of a known pattern on either side of `buf` (the part the frame and the tests' prefills use); download() checks them, so a
    kernel that writes outside the destination picture it was given fails the conversion that did it, in every GPU test, instead of damaging whatever torch placed
    next to the frame (hunt for the writer, not the victim)."""
    GUARD = 1024
    GUARD_BYTE = 0xA5

    def __init__(self, fmt, w, h, device="cuda:0"):
        import torch
        self.fmt, self.w, self.h = fmt, w, h
        self.linesize, self.offset, self.total = image_layout(fmt, w, h, 256)
        self._allocate(self.total + 256, device)
        base = self.buf.data_ptr()
        self.base = (base + 255) // 256 * 256
        self._shift = self.base - base
        self.nplanes = len(plane_layout(fmt, w, h))
        self.row_bytes = [rb for rb, _ in plane_layout(fmt, w, h)]
        self.rows = [r for _, r in plane_layout(fmt, w, h)]

    def ptrs(self):
        p = (C.c_void_p * 4)()
        s = (C.c_int * 4)()
        for i in range(self.nplanes):
            p[i] = self.base + self.offset[i]
            s[i] = self.linesize[i]
        return p, s

    def view(self):
        v = SwsFrameView()
        for i in range(self.nplanes):
            v.data[i] = self.base + self.offset[i]
            v.linesize[i] = self.linesize[i]
        v.width, v.height, v.format = self.w, self.h, PIX_FMT[self.fmt]
        return apply_props(v, getattr(self, "props", None))

    def plane_tensor(self, i):
        """2-D (rows, linesize) uint8 view of plane i."""
        o = self._shift + self.offset[i]
        return self.buf[o:o + self.rows[i] * self.linesize[i]].view(self.rows[i], self.linesize[i])

    def upload(self, host):
        import torch
        for i, a in enumerate(host.planes):
            rb = self.row_bytes[i]
            self.plane_tensor(i)[:, :rb].copy_(torch.from_numpy(np.ascontiguousarray(a[:, :rb])))
        torch.cuda.synchronize()  # the context runs on its own stream: make the upload visible to it
        return self

    def _allocate(self, nbytes, device):
        """`buf` = nbytes zeroed bytes between two guard bands of one allocation"""
        import torch
        G = self.GUARD
        self._alloc = torch.full((G + nbytes + G,), self.GUARD_BYTE, dtype=torch.uint8, device=device)
        self.buf = self._alloc[G:G + nbytes]
        self.buf.zero_()

    def check_guards(self):
        """raises if anything wrote into the guard bytes around the frame (host-synchronous)"""
        G = self.GUARD
        if getattr(self, "_alloc", None) is None or self.buf.data_ptr() != self._alloc.data_ptr() + G:
            return      # a frame whose storage a test replaced with its own
        lo, hi = self._alloc[:G], self._alloc[G + self.buf.numel():]
        bad_lo, bad_hi = int((lo != self.GUARD_BYTE).sum()), int((hi != self.GUARD_BYTE).sum())
        if bad_lo or bad_hi:
            raise AssertionError(f"{self.fmt} {self.w}x{self.h} frame at {self.base:#x}: {bad_lo} guard bytes BEFORE and {bad_hi} AFTER the frame were overwritten "
                                 f"(something wrote outside the picture it was given)")

    def download(self, host=None):
        host = host or HostFrame(self.fmt, self.w, self.h)
        for i, a in enumerate(host.planes):
            rb = self.row_bytes[i]
            a[:, :rb] = self.plane_tensor(i)[:, :rb].cpu().numpy()
        self.check_guards()
        return host


class SwsContextFields(C.Structure):
    """public part of SwsContext (include/swscale_hip.h; reference swscale.h:227-315), for the
    sws_alloc_context() + set fields + sws_init_context() construction the reference documents."""
    _fields_ = [("av_class", C.c_void_p), ("opaque", C.c_void_p), ("flags", C.c_uint), ("scaler_params", C.c_double * 2),
                ("threads", C.c_int), ("dither", C.c_int), ("alpha_blend", C.c_int), ("gamma_flag", C.c_int),
                ("src_w", C.c_int), ("src_h", C.c_int), ("dst_w", C.c_int), ("dst_h", C.c_int),
                ("src_format", C.c_int), ("dst_format", C.c_int), ("src_range", C.c_int), ("dst_range", C.c_int),
                ("src_v_chr_pos", C.c_int), ("src_h_chr_pos", C.c_int), ("dst_v_chr_pos", C.c_int), ("dst_h_chr_pos", C.c_int),
                ("intent", C.c_int), ("scaler", C.c_int), ("scaler_sub", C.c_int), ("backends", C.c_int)]


class SwsContext:
    """sws_getContext(...) wrapper.  scale() takes HostFrame or DeviceFrame objects.
    Keyword options (dither=, src_range=, dst_range=, src_h_chr_pos=, ...) select the sws_alloc_context() + public
    fields + sws_init_context() construction instead."""

    def __init__(self, sw, sh, sfmt, dw, dh, dfmt, flags, param=None, device=None, empty=False, **opts):
        L = load_library()
        self.L = L
        self.sw, self.sh, self.sfmt, self.dw, self.dh, self.dfmt = sw, sh, sfmt, dw, dh, dfmt
        if empty:
            self.c = L.sws_alloc_context()
        elif opts:
            opts = dict(opts)
            self.c = L.sws_alloc_context()
            if not self.c:
                raise RuntimeError("sws_alloc_context failed")
            f = self.fields()
            f.src_w, f.src_h, f.src_format = sw, sh, PIX_FMT[sfmt]
            f.dst_w, f.dst_h, f.dst_format = dw, dh, PIX_FMT[dfmt]
            f.flags = flags
            if param:
                f.scaler_params[0], f.scaler_params[1] = param
            src_filter = opts.pop("src_filter", None)     # POINTER(SwsFilter) from sws_getDefaultFilter(), or None
            dst_filter = opts.pop("dst_filter", None)
            for k, v in opts.items():
                setattr(f, k, v)
            r = L.sws_init_context(self.c, src_filter, dst_filter)
            if r < 0:
                L.sws_freeContext(self.c)
                self.c = None
                raise RuntimeError(f"sws_init_context({sfmt} -> {dfmt}, {opts}) = {r}")
        else:
            p = (C.c_double * 2)(*param) if param else None
            self.c = L.sws_getContext(sw, sh, PIX_FMT[sfmt], dw, dh, PIX_FMT[dfmt], flags, None, None, p)
        if not self.c:
            raise RuntimeError(f"sws_getContext({sfmt} {sw}x{sh} -> {dfmt} {dw}x{dh}, flags={flags:#x}) failed")
        if device is not None:
            r = L.sws_hip_set_device(self.c, int(device))
            if r < 0:
                raise RuntimeError(f"sws_hip_set_device({device}) = {r}")

    def fields(self):
        return C.cast(self.c, C.POINTER(SwsContextFields)).contents

    def set_colorspace(self, inv_cs, src_range, cs, dst_range, brightness=0, contrast=1 << 16, saturation=1 << 16):
        L = self.L
        inv = (C.c_int * 4)(*[L.sws_getCoefficients(inv_cs)[i] for i in range(4)])
        tab = (C.c_int * 4)(*[L.sws_getCoefficients(cs)[i] for i in range(4)])
        return L.sws_setColorspaceDetails(self.c, inv, src_range, tab, dst_range, brightness, contrast, saturation)

    def scale(self, src, dst, slice_y=0, slice_h=None):
        sp, ss = src.ptrs()
        dp, ds = dst.ptrs()
        return self.L.sws_scale(self.c, sp, ss, slice_y, self.fields().src_h if slice_h is None else slice_h, dp, ds)

    def scale_frame(self, src, dst):
        """sws_scale_frame(); on an sws_alloc_context()ed context (empty=True) the library configures itself from the frames."""
        sv, dv = src.view(), dst.view()
        return self.L.sws_scale_frame(self.c, C.byref(dv), C.byref(sv))

    def scale_frames(self, srcs, dsts):
        n = len(srcs)
        sv = [s.view() for s in srcs]
        dv = [d.view() for d in dsts]
        sa = (C.POINTER(SwsFrameView) * n)(*[C.pointer(v) for v in sv])
        da = (C.POINTER(SwsFrameView) * n)(*[C.pointer(v) for v in dv])
        return self.L.sws_scale_frames(self.c, da, sa, n)

    def make_batch(self, srcs, dsts):
        """Pre-build the ctypes arrays for repeated sws_scale_frames() calls (bench inner loop)."""
        n = len(srcs)
        sv = [s.view() for s in srcs]
        dv = [d.view() for d in dsts]
        sa = (C.POINTER(SwsFrameView) * n)(*[C.pointer(v) for v in sv])
        da = (C.POINTER(SwsFrameView) * n)(*[C.pointer(v) for v in dv])
        return (sa, da, n, sv, dv)

    def run_batch(self, batch):
        return self.L.sws_scale_frames(self.c, batch[1], batch[0], batch[2])

    def set_stream(self, stream_handle):
        return self.L.sws_hip_set_stream(self.c, C.c_void_p(stream_handle))

    def sync(self):
        return self.L.sws_hip_sync(self.c)

    def set_option(self, name, value):
        """launch heuristics of the context (sws_hip_set_option): e.g. strip_min_w, max_devices, no_strip"""
        r = self.L.sws_hip_set_option(self.c, name.encode(), int(value))
        if r < 0:
            raise ValueError(name)
        return r

    def set_timing(self, on=True):
        return self.L.sws_hip_set_timing(self.c, 1 if on else 0)

    def last_kernel_ms(self):
        return self.L.sws_hip_last_kernel_ms(self.c)

    def path(self):
        self.L.sws_hip_get_stream(self.c)  # forces dev_prepare so the path is named
        return self.L.sws_hip_path_name(self.c).decode()

    def kernel_name(self):
        return self.L.sws_hip_kernel_name(self.c).decode()

    def debug_check(self):
        """(anomalies, text): the context's device tables read back and compared with what was uploaded (sws_hip_debug_check)"""
        buf = C.create_string_buffer(2048)
        n = self.L.sws_hip_debug_check(self.c, buf, len(buf))
        return n, buf.value.decode(errors="replace")

    def filter(self, which):
        f = C.POINTER(C.c_int16)()
        p = C.POINTER(C.c_int32)()
        n = C.c_int()
        fs = self.L.sws_hip_get_filter(self.c, which, C.byref(f), C.byref(p), C.byref(n))
        if not fs:
            return 0, None, None
        return fs, np.ctypeslib.as_array(f, shape=(n.value, fs)).copy(), np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def tables(self):
        r2y = (C.c_int32 * 9)()
        y2r = (C.c_int * 6)()
        co = (C.c_uint32 * 2)()
        of = (C.c_int64 * 2)()
        act = self.L.sws_hip_get_tables(self.c, r2y, y2r, co, of)
        return list(r2y), list(y2r), list(co), list(of), act

    def export_tables(self):
        n = self.L.sws_hip_tables_size(self.c)
        buf = (C.c_uint8 * n)()
        r = self.L.sws_hip_tables_export(self.c, buf, n)
        if r < 0:
            raise RuntimeError("tables export failed")
        return bytes(buf)

    def import_tables(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        r = self.L.sws_hip_tables_import(self.c, buf, len(blob))
        if r < 0:
            raise RuntimeError("tables import failed")
        self._refresh_geometry()
        return r

    def _refresh_geometry(self):
        """the wrapper's copy of the geometry follows the C struct (a context that was configured by import_tables())"""
        f = self.fields()
        self.sw, self.sh, self.dw, self.dh = f.src_w, f.src_h, f.dst_w, f.dst_h
        names = {v: k for k, v in PIX_FMT.items()}
        self.sfmt, self.dfmt = names.get(f.src_format, self.sfmt), names.get(f.dst_format, self.dfmt)

    def close(self):
        if getattr(self, "c", None):
            self.L.sws_freeContext(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
