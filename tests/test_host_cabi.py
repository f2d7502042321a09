"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol declared in
include/swscale_hip.h, and its host-side init (filters, colour tables, path selection) agrees with the oracle.
No compute calls: there is no GPU here and the product has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from test_oracle_golden import GOLD, flags_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "swscale_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:sws|swscale)_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35
    missing = [n for n in sorted(names) if not hasattr(hiplib, n)]
    assert not missing, missing
    assert hiplib.swscale_version() == (10 << 16) | (2 << 8) | 100


def test_swscontext_public_layout(hiplib):
    """SwsContext is the first member of the private struct and has the reference's field order (swscale.h:227-315)."""
    class Pub(C.Structure):
        _fields_ = [("av_class", C.c_void_p), ("opaque", C.c_void_p), ("flags", C.c_uint), ("scaler_params", C.c_double * 2),
                    ("threads", C.c_int), ("dither", C.c_int), ("alpha_blend", C.c_int), ("gamma_flag", C.c_int),
                    ("src_w", C.c_int), ("src_h", C.c_int), ("dst_w", C.c_int), ("dst_h", C.c_int),
                    ("src_format", C.c_int), ("dst_format", C.c_int), ("src_range", C.c_int), ("dst_range", C.c_int),
                    ("src_v_chr_pos", C.c_int), ("src_h_chr_pos", C.c_int), ("dst_v_chr_pos", C.c_int), ("dst_h_chr_pos", C.c_int),
                    ("intent", C.c_int), ("scaler", C.c_int), ("scaler_sub", C.c_int), ("backends", C.c_int)]
    c = LA.SwsContext(1920, 1080, "nv12", 1280, 720, "bgr0", LA.SWS_LANCZOS | LA.SWS_BITEXACT)
    p = C.cast(c.c, C.POINTER(Pub)).contents
    assert (p.src_w, p.src_h, p.dst_w, p.dst_h) == (1920, 1080, 1280, 720)
    assert p.src_format == LA.PIX_FMT["nv12"] and p.dst_format == LA.PIX_FMT["bgra"]  # bgr0 -> bgra (handle_0alpha)
    assert p.flags == LA.SWS_LANCZOS | LA.SWS_BITEXACT and p.dither == 1 and p.threads == 1
    assert p.src_h_chr_pos == -513 and p.scaler_params[0] == 123456


CASES = [
    (1280, 720, "yuv420p", 640, 360, "yuv420p", "SWS_BILINEAR|SWS_BITEXACT"),
    (3840, 2160, "yuv420p", 3840, 2160, "rgb24", "SWS_BICUBIC|SWS_BITEXACT|SWS_ACCURATE_RND"),
    (7680, 4320, "yuv420p10le", 3840, 2160, "p010le", "SWS_LANCZOS|SWS_BITEXACT"),
    (3840, 2160, "yuv420p10le", 7680, 4320, "p010le", "SWS_LANCZOS|SWS_BITEXACT"),
    (1920, 1080, "nv12", 1920, 1080, "bgr0", "SWS_BICUBIC|SWS_BITEXACT"),
    (3840, 2160, "gbrpf32le", 3840, 2160, "yuv444p16le", "SWS_BICUBIC|SWS_BITEXACT"),
    (352, 288, "rgb24", 200, 100, "yuv420p", "SWS_BICUBIC|SWS_BITEXACT|SWS_ACCURATE_RND"),
    (352, 288, "bgra", 200, 100, "nv12", "SWS_BICUBIC"),
    (641, 479, "yuv420p", 1001, 777, "yuv444p", "SWS_LANCZOS"),
    (640, 480, "yuv420p", 333, 101, "nv12", "SWS_AREA"),
    (640, 480, "yuv422p", 1333, 1001, "nv12", "SWS_AREA"),
    (640, 480, "yuv420p", 333, 101, "yuv444p", "SWS_GAUSS"),
    (640, 480, "yuv420p", 700, 501, "yuv444p", "SWS_SINC"),
    (640, 480, "yuv420p", 700, 501, "yuv444p", "SWS_SPLINE"),
    (640, 480, "yuv420p", 700, 501, "yuv444p", "SWS_POINT"),
    (640, 480, "yuv420p", 300, 201, "yuv444p", "SWS_X"),
    (640, 480, "yuv420p", 300, 201, "yuv422p", "SWS_BICUBLIN"),
    (64, 48, "yuv444p", 3, 5, "rgb24", "SWS_BICUBIC"),
    (5, 3, "yuv420p", 640, 480, "rgba", "SWS_BILINEAR"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]}")
def test_host_tables_match_oracle(hiplib, case):
    sw, sh, sf, dw, dh, df, fl = case
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags_of(fl))
    p = LA.SwsContext(sw, sh, sf, dw, dh, df, flags_of(fl))
    for which in range(4):
        a, b = o.filter(which), p.filter(which)
        assert a[0] == b[0]
        if a[0]:
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    r2y, y2r, co, of, act = p.tables()
    assert r2y == o.rgb2yuv()
    if df in ("rgb24", "rgba", "bgr0"):
        assert y2r == o.yuv2rgb_coeffs()
    oc, oo, oa = o.range_consts()
    assert act == oa and (not act or (co == oc and of == oo))


def test_colorspace_details_and_range_constants(hiplib):
    g = GOLD["c5"]
    sw, sh, sf, dw, dh, df, fl = g["ctx"]
    p = LA.SwsContext(sw, sh, sf, dw, dh, df, flags_of(fl))
    assert p.set_colorspace(*g["colorspace"]) == 0
    r2y, _, co, of, act = p.tables()
    assert r2y == g["rgb2yuv"] and act == 1
    assert co == [g["range"]["lumCoeff"], g["range"]["chrCoeff"]] and of == [g["range"]["lumOffset"], g["range"]["chrOffset"]]


def test_unsupported_requests_fail_like_the_reference(hiplib):
    L = hiplib
    assert not L.sws_getContext(0, 10, 0, 10, 10, 0, 4, None, None, None)          # invalid dimension -> NULL
    assert not L.sws_getContext(16, 16, 0, 16, 16, 0, 4 | 2, None, None, None)     # two scaler flags -> NULL
    assert not L.sws_getContext(16, 16, 0, 16, 16, 11, 4, None, None, None)        # pal8 is an input only, like in the reference
    assert not L.sws_getContext(16, 16, 0, 16, 16, 139, 4, None, None, None)       # bayer_bggr8 too
    assert not L.sws_getContext(16, 16, 16000, 16, 16, 0, 4, None, None, None)     # not a pixel format
    assert L.sws_isSupportedInput(LA.PIX_FMT["nv12"]) and L.sws_isSupportedOutput(LA.PIX_FMT["p010le"])
    assert L.sws_isSupportedOutput(LA.PIX_FMT["gbrpf32le"]) and L.sws_isSupportedInput(LA.PIX_FMT["gbrp12le"])
    assert L.sws_isSupportedOutput(LA.PIX_FMT["gray8"]) and L.sws_isSupportedInput(LA.PIX_FMT["gray12le"])
    assert L.sws_isSupportedInput(9) and L.sws_isSupportedInput(11) and not L.sws_isSupportedOutput(11)   # monowhite; pal8 is read, never written
    L.sws_freeContext(None)  # NULL-safe (swscale.h:528)
    c = LA.SwsContext(64, 64, "yuv420p", 32, 32, "yuv420p", LA.SWS_BICUBIC)
    sp = (C.c_void_p * 4)()
    ss = (C.c_int * 4)()
    assert L.sws_scale(c.c, None, ss, 0, 64, sp, ss) == -22            # NULL arguments -> AVERROR(EINVAL)
    assert L.sws_scale(c.c, sp, ss, 1, 8, sp, ss) == -22               # slice not aligned to chroma rows
    assert L.sws_scale(c.c, sp, ss, 0, 64, sp, ss) == -22              # bad plane pointers


def test_table_blob_roundtrip(hiplib):
    """what rank 0 broadcasts to the other GPUs: export -> import into an sws_alloc_context() shell."""
    a = LA.SwsContext(1920, 1080, "nv12", 1280, 720, "bgr0", LA.SWS_LANCZOS | LA.SWS_BITEXACT)
    blob = a.export_tables()
    b = LA.SwsContext(0, 0, "nv12", 0, 0, "bgr0", 0, empty=True)
    b.import_tables(blob)
    for which in range(4):
        x, y = a.filter(which), b.filter(which)
        assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])
    assert a.tables() == b.tables()
    assert b.export_tables() == blob
    # a cascaded context (YUV matrix change) carries its two children
    c = LA.SwsContext(96, 64, "yuv420p", 96, 64, "yuv420p", LA.SWS_BICUBIC | LA.SWS_ACCURATE_RND)
    assert c.set_colorspace(LA.SWS_CS_ITU709, 0, LA.SWS_CS_ITU601, 1) == 0
    d = LA.SwsContext(0, 0, "yuv420p", 0, 0, "yuv420p", 0, empty=True)
    d.import_tables(c.export_tables())
    assert d.export_tables() == c.export_tables()


def test_prefixed_twin_exports_the_same_api_as_swship(hiplib):
    """libswship.so (make prefixed): every export of libswscale_hip.so under the swship_ prefix and nothing else, so that a process can
    hold it next to the real libswscale; include/swscale_hip_prefix.h is the rename list."""
    import subprocess
    libdir = os.path.join(ROOT, "librempeg_amd", "lib")
    def exports(name):
        out = subprocess.check_output(["nm", "-D", os.path.join(libdir, name)], text=True)
        return sorted(l.split()[2].split("@")[0] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")
    plain, pref = exports("libswscale_hip.so"), exports("libswship.so")
    assert all(s.startswith(("sws_", "swscale_")) for s in plain) and all(s.startswith("swship_") for s in pref)
    ren = lambda s: "swship_" + s[4:] if s.startswith("sws_") else "swship_" + s
    assert sorted(ren(s) for s in plain) == pref
    hdr = open(os.path.join(ROOT, "include", "swscale_hip_prefix.h")).read()
    assert sorted(re.findall(r"^#define (\w+) swship_\w+$", hdr, flags=re.M)) == plain
    both = C.CDLL(os.path.join(libdir, "libswship.so"))
    assert both.swship_swscale_version() == hiplib.swscale_version()
    both.swship_alloc_context.restype = C.c_void_p
    c = both.swship_alloc_context()
    assert c
    both.swship_freeContext.argtypes = [C.c_void_p]
    both.swship_freeContext(c)


def test_format_queries_answer_like_the_reference_table(hiplib):
    """sws_isSupportedInput() / sws_isSupportedOutput() for every AVPixelFormat value against libswscale/format.c legacy_format_entries
    (tests/golden/legacy_format_entries.json, written by tools/gen_format_table.py): all 234 rows, and nothing outside the table."""
    import json
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legacy_format_entries.json")))
    L = hiplib
    want = {e["value"]: (e["in"], e["out"]) for e in tab["entries"]}
    assert len(want) == 234
    for v in range(-1, tab["nb"] + 8):
        assert (L.sws_isSupportedInput(v), L.sws_isSupportedOutput(v)) == want.get(v, (0, 0)), v
