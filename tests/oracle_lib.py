"""ctypes binding of the CPU oracle (oracle/libsws_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import re
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# The format table (AVPixelFormat value, plane layout, chroma shifts, bytes per sample) and plane_layout() are the product's own descriptions of the
# caller-visible picture layout (librempeg_amd/swscale.py): one copy, imported here -- they describe buffers, not arithmetic, and the reference's
# goldens (MD5s over these very buffers) pin them.
import sys as _sys
if ROOT not in _sys.path:
    _sys.path.insert(0, ROOT)
from librempeg_amd.swscale import _FORMATS, plane_layout  # noqa: E402

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
