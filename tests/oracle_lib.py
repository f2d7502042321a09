"""ctypes binding of the CPU oracle (oracle/libsws_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import re
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# AVPixelFormat values (libavutil/pixfmt.h enum order)
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
FMT = {k: v[0] for k, v in _FORMATS.items()}

SWS_FAST_BILINEAR, SWS_BILINEAR, SWS_BICUBIC, SWS_X, SWS_POINT, SWS_AREA = 1, 2, 4, 8, 16, 32
SWS_BICUBLIN, SWS_GAUSS, SWS_SINC, SWS_LANCZOS, SWS_SPLINE = 64, 128, 256, 512, 1024
SWS_PRINT_INFO, SWS_FULL_CHR_H_INT, SWS_FULL_CHR_H_INP = 1 << 12, 1 << 13, 1 << 14
SWS_ACCURATE_RND, SWS_BITEXACT = 1 << 18, 1 << 19
SWS_CS_ITU709, SWS_CS_ITU601, SWS_CS_DEFAULT, SWS_CS_BT2020 = 1, 5, 5, 9


class Frame:
    """Host frame: list of 2-D uint8 numpy planes with 64-byte aligned strides."""

    def __init__(self, fmt, w, h, align=64, fill=None):
        self.fmt, self.w, self.h = fmt, w, h
        self.planes = []
        for (rb, rows) in plane_layout(fmt, w, h):
            stride = (rb + align - 1) // align * align if align else rb
            a = np.zeros((rows, stride), dtype=np.uint8)
            if fill is not None:
                a[:] = fill
            self.planes.append(a)
        self.row_bytes = [rb for rb, _ in plane_layout(fmt, w, h)]

    def ptrs(self):
        p = (C.c_void_p * 4)()
        s = (C.c_int * 4)()
        for i, a in enumerate(self.planes):
            p[i] = a.ctypes.data
            s[i] = a.strides[0]
        return p, s

    def visible(self):
        """concatenated visible bytes of all planes (what framecrc/rawvideo would see)."""
        return b"".join(a[:, :rb].tobytes() for a, rb in zip(self.planes, self.row_bytes))

    def visible_arrays(self):
        return [a[:, :rb] for a, rb in zip(self.planes, self.row_bytes)]


def xorshift_bytes(n_u64, seed):
    """xorshift64* stream, state0 = 0x9E3779B97F4A7C15 ^ seed -> uint64 array."""
    out = np.empty(n_u64, dtype=np.uint64)
    # vectorised jump-free generation is awkward; use a simple blocked python loop on numpy scalars
    x = np.uint64(0x9E3779B97F4A7C15 ^ seed)
    m = np.uint64(0x2545F4914F6CDD1D)
    with np.errstate(over="ignore"):
        for i in range(n_u64):
            x ^= x >> np.uint64(12)
            x ^= x << np.uint64(25)
            x ^= x >> np.uint64(27)
            out[i] = x * m
    return out


def fill_random(frame, seed):
    """Deterministic pseudo-random content, format aware (10-bit masks, floats in [0,1) with outliers)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f = frame.fmt
    for pi, (a, rb) in enumerate(zip(frame.planes, frame.row_bytes)):
        rows = a.shape[0]
        m = re.match(r"(?:yuva?4\d\dp|gbra?p|gray)(9|10|12|14)[lb]e$", f)
        depth = 10 if f in ("nv20le", "nv20be") else int(m.group(1)) if m else 0
        mp = re.match(r"(?:p[024]|yuv444p|gbrp)(10|12)(?:msb)?[lb]e$", f) if ("msb" in f or f[0] == "p") else None
        be = f.endswith("be")
        if depth:  # N-bit samples in the low bits of 16-bit words
            v = rng.integers(0, 1 << depth, size=(rows, rb // 2), dtype=np.uint16)
            a[:, :rb] = (v.byteswap() if be else v).view(np.uint8).reshape(rows, rb)
        elif mp:   # N-bit samples in the high bits (p010 / p012 families)
            d = int(mp.group(1))
            v = (rng.integers(0, 1 << d, size=(rows, rb // 2), dtype=np.uint16) << (16 - d)).astype(np.uint16)
            a[:, :rb] = (v.byteswap() if be else v).view(np.uint8).reshape(rows, rb)
        elif f[:-2] in ("rgbf16", "rgbaf16", "grayf16", "yaf16", "gbrpf16", "gbrapf16"):
            v = rng.random(size=(rows, rb // 2), dtype=np.float32).astype(np.float16)
            flat = v.reshape(-1)
            flat[::257] = -0.25
            flat[128::257] = 1.25
            bits = flat.view(np.uint16)
            bits[64::131] = rng.integers(0, 65536, size=len(bits[64::131]), dtype=np.uint16)   # any bit pattern: subnormals, infinities, NaNs
            a[:, :rb] = (v.byteswap() if be else v).view(np.uint8).reshape(rows, rb)
        elif f in ("gbrpf32le", "gbrpf32be", "grayf32le", "grayf32be", "gbrapf32le", "gbrapf32be", "rgbf32le", "rgbf32be", "yaf32le", "yaf32be"):
            v = rng.random(size=(rows, rb // 4), dtype=np.float32)
            flat = v.reshape(-1)
            flat[::257] = -0.25
            flat[128::257] = 1.25
            flat[64::1031] = np.nan; flat[65::1031] = np.inf; flat[66::1031] = -np.inf; flat[67::1031] = 1e-42   # av_clipf's NaN rule, a denormal
            a[:, :rb] = (v.byteswap() if be else v).view(np.uint8).reshape(rows, rb)
        else:
            a[:, :rb] = rng.integers(0, 256, size=(rows, rb), dtype=np.uint8)
    return frame


class OrSwsOpts(C.Structure):
    _fields_ = [("src_w", C.c_int), ("src_h", C.c_int), ("src_format", C.c_int),
                ("dst_w", C.c_int), ("dst_h", C.c_int), ("dst_format", C.c_int),
                ("flags", C.c_uint), ("scaler_params", C.c_double * 2), ("dither", C.c_int),
                ("src_range", C.c_int), ("dst_range", C.c_int),
                ("src_v_chr_pos", C.c_int), ("src_h_chr_pos", C.c_int),
                ("dst_v_chr_pos", C.c_int), ("dst_h_chr_pos", C.c_int),
                ("src_vec", C.POINTER(C.c_double) * 4), ("src_vec_len", C.c_int * 4), ("dst_vec_len", C.c_int * 4), ("gamma_flag", C.c_int), ("alpha_blend", C.c_int)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("SWS_ORACLE_LIBRARY")          # tools/asan_env.sh: the sanitizer build of the checker
        if so:
            if not os.path.exists(so):
                raise RuntimeError(f"SWS_ORACLE_LIBRARY={so} does not exist (make -C oracle asan)")
        else:
            so = os.path.join(ORACLE_DIR, "libsws_oracle.so")
        if "SWS_ORACLE_LIBRARY" in os.environ:
            pass
        elif not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, "sws_oracle.c")):
            build()
        L = C.CDLL(so)
        L.or_sws_get_context.restype = C.c_void_p
        L.or_sws_get_context.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
        L.or_sws_create.restype = C.c_void_p
        L.or_sws_create.argtypes = [C.POINTER(OrSwsOpts)]
        L.or_sws_default_opts.argtypes = [C.POINTER(OrSwsOpts)]
        L.or_sws_set_colorspace.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.c_int, C.c_int, C.c_int]
        L.or_sws_get_coefficients.restype = C.POINTER(C.c_int)
        L.or_sws_get_coefficients.argtypes = [C.c_int]
        L.or_sws_scale.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.or_sws_free.argtypes = [C.c_void_p]
        L.or_sws_get_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_int16)),
                                        C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int)]
        L.or_sws_path.argtypes = [C.c_void_p]
        L.or_sws_path_name.argtypes = [C.c_void_p]
        L.or_sws_path_name.restype = C.c_char_p
        L.or_sws_rgb2yuv_table.argtypes = [C.c_void_p]
        L.or_sws_rgb2yuv_table.restype = C.POINTER(C.c_int32)
        L.or_sws_yuv2rgb_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.or_sws_range_consts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.or_sws_lut_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.or_sws_lut_rgb.restype = C.c_uint32
        L.or_sws_chroma_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


class Oracle:
    """Thin OO wrapper: Oracle(srcW,srcH,'yuv420p',dstW,dstH,'rgb24',flags)."""

    def __init__(self, sw, sh, sfmt, dw, dh, dfmt, flags, param=None, **opts):
        L = lib()
        self.sfmt, self.dfmt, self.sw, self.sh, self.dw, self.dh = sfmt, dfmt, sw, sh, dw, dh
        if opts:
            o = OrSwsOpts()
            L.or_sws_default_opts(C.byref(o))
            o.src_w, o.src_h, o.src_format = sw, sh, FMT[sfmt]
            o.dst_w, o.dst_h, o.dst_format = dw, dh, FMT[dfmt]
            o.flags = flags
            if param:
                o.scaler_params[0], o.scaler_params[1] = param
            self._vecs = []
            for k, v in opts.items():
                if k == "src_filter":      # {"lumH": [...], "lumV": [...], "chrH": [...], "chrV": [...]}
                    for idx, name in enumerate(("lumH", "lumV", "chrH", "chrV")):
                        if v.get(name) is not None:
                            arr = (C.c_double * len(v[name]))(*v[name])
                            self._vecs.append(arr)
                            o.src_vec[idx] = C.cast(arr, C.POINTER(C.c_double))
                            o.src_vec_len[idx] = len(v[name])
                elif k == "dst_filter_len":
                    for idx, name in enumerate(("lumH", "lumV", "chrH", "chrV")):
                        o.dst_vec_len[idx] = int(v.get(name, 0))
                else:
                    setattr(o, k, v)
            self.c = L.or_sws_create(C.byref(o))
        else:
            p = (C.c_double * 2)(*param) if param else None
            self.c = L.or_sws_get_context(sw, sh, FMT[sfmt], dw, dh, FMT[dfmt], flags, p)
        if not self.c:
            raise RuntimeError(f"oracle: unsupported {sfmt}->{dfmt}")

    def set_colorspace(self, inv_cs, src_range, cs, dst_range, brightness=0, contrast=1 << 16, saturation=1 << 16):
        L = lib()
        inv = (C.c_int * 4)(*[L.or_sws_get_coefficients(inv_cs)[i] for i in range(4)])
        tab = (C.c_int * 4)(*[L.or_sws_get_coefficients(cs)[i] for i in range(4)])
        return L.or_sws_set_colorspace(self.c, inv, src_range, tab, dst_range, brightness, contrast, saturation)

    def scale(self, src, dst):
        sp, ss = src.ptrs()
        dp, dstr = dst.ptrs()
        return lib().or_sws_scale(self.c, sp, ss, 0, self.sh, dp, dstr)

    def filter(self, which):
        f = C.POINTER(C.c_int16)()
        p = C.POINTER(C.c_int32)()
        n = C.c_int()
        fs = lib().or_sws_get_filter(self.c, which, C.byref(f), C.byref(p), C.byref(n))
        if not fs or not f:
            return 0, None, None
        taps = np.ctypeslib.as_array(f, shape=(n.value, fs)).copy()
        pos = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        return fs, taps, pos

    def path(self):
        return lib().or_sws_path_name(self.c).decode()

    def rgb2yuv(self):
        t = lib().or_sws_rgb2yuv_table(self.c)
        return [t[i] for i in range(9)]

    def yuv2rgb_coeffs(self):
        o = (C.c_int * 6)()
        lib().or_sws_yuv2rgb_coeffs(self.c, o)
        return list(o)

    def range_consts(self):
        co = (C.c_uint32 * 2)()
        of = (C.c_int64 * 2)()
        a = C.c_int()
        lib().or_sws_range_consts(self.c, co, of, C.byref(a))
        return list(co), list(of), a.value

    def chroma_dims(self):
        o = (C.c_int * 8)()
        lib().or_sws_chroma_dims(self.c, o)
        return list(o)

    def close(self):
        if self.c:
            lib().or_sws_free(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
