"""CPU checks of the host-only part of the public API (vectors, SwsFilter, palette helpers, capability tests) and of
SwsFilter handling in the filter builder (product host init vs oracle).  No compute calls."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as OL
import librempeg_amd as LA
from librempeg_amd import swscale as S


def gaussian(variance, quality):
    """closed form of sws_getGaussianVec (reference utils.c:1968-2001)"""
    n = int(variance * quality + 0.5) | 1
    mid = (n - 1) * 0.5
    v = np.array([math.exp(-(i - mid) ** 2 / (2 * variance * variance)) / math.sqrt(2 * variance * math.pi) for i in range(n)])
    return v / v.sum()


def test_vector_api(hiplib):
    L = hiplib
    assert not L.sws_allocVec(0) and not L.sws_allocVec(-3)
    v = L.sws_allocVec(5)
    assert v and v.contents.length == 5
    for i in range(5):
        v.contents.coeff[i] = i + 1
    L.sws_scaleVec(v, 2.0)
    assert S.vec_to_list(v) == [2, 4, 6, 8, 10]
    L.sws_normalizeVec(v, 1.0)
    assert abs(sum(S.vec_to_list(v)) - 1.0) < 1e-12
    L.sws_freeVec(v)
    L.sws_freeVec(None)
    for var, q in ((1.0, 3.0), (2.5, 3.0), (0.4, 3.0), (3.0, 1.0)):
        g = L.sws_getGaussianVec(var, q)
        assert np.allclose(S.vec_to_list(g), gaussian(var, q), rtol=0, atol=1e-15)
        L.sws_freeVec(g)
    assert not L.sws_getGaussianVec(-1.0, 3.0)


def test_default_filter(hiplib):
    L = hiplib
    f = L.sws_getDefaultFilter(0, 0, 0, 0, 0, 0, 0)
    assert S.filter_to_dict(f) == {"lumH": [1.0], "lumV": [1.0], "chrH": [1.0], "chrV": [1.0]}
    L.sws_freeFilter(f)
    f = L.sws_getDefaultFilter(1.5, 2.0, 0.5, 0.25, 1.0, 0.0, 0)
    d = S.filter_to_dict(f)
    L.sws_freeFilter(f)
    L.sws_freeFilter(None)
    assert not L.sws_getDefaultFilter(0, 0, 1.0, 0, 0, 0, 0)    # identity - identity: normalising 0 gives NaN -> NULL (utils.c:1948-1952)
    # luma: -0.5 * gaussian + identity at the centre, normalised to 1
    g = -0.5 * gaussian(1.5, 3.0)
    g[len(g) // 2] += 1.0
    g /= g.sum()
    assert np.allclose(d["lumH"], g, atol=1e-15) and d["lumH"] == d["lumV"]
    # chroma: sharpened gaussian shifted one sample to the left inside a window two taps wider
    c = -0.25 * gaussian(2.0, 3.0)
    c[len(c) // 2] += 1.0
    c /= c.sum()
    assert np.allclose(d["chrV"], c, atol=1e-15)
    assert len(d["chrH"]) == len(c) + 2 and np.allclose(d["chrH"][:len(c)], c, atol=1e-15) and d["chrH"][-2:] == [0.0, 0.0]


def test_palette_helpers(hiplib):
    rng = np.random.default_rng(5)
    pal = rng.integers(0, 256, 1024, dtype=np.uint8)
    src = rng.integers(0, 256, 777, dtype=np.uint8)
    d32 = np.zeros(777 * 4, np.uint8)
    d24 = np.zeros(777 * 3, np.uint8)
    hiplib.sws_convertPalette8ToPacked32(src.ctypes.data, d32.ctypes.data, 777, pal.ctypes.data)
    hiplib.sws_convertPalette8ToPacked24(src.ctypes.data, d24.ctypes.data, 777, pal.ctypes.data)
    p = pal.reshape(256, 4)
    assert np.array_equal(d32.reshape(-1, 4), p[src]) and np.array_equal(d24.reshape(-1, 3), p[src][:, :3])


def test_capability_queries(hiplib):
    L = hiplib
    P = LA.PIX_FMT
    assert L.sws_get_class() and C.cast(L.sws_get_class(), C.POINTER(C.c_char_p))[0] == b"SWScaler"
    assert L.sws_test_format(P["yuv420p"], 0) and L.sws_test_format(P["rgb24"], 1)
    assert not L.sws_test_format(-1, 0) and not L.sws_test_format(11, 1)
    assert L.sws_test_hw_format(-1) and L.sws_test_hw_format(268) and not L.sws_test_hw_format(P["yuv420p"])
    assert [L.sws_test_colorspace(i, 0) for i in range(12)] == [1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 0, 0]   # format.c:625-640
    assert [L.sws_test_primaries(i, 0) for i in (0, 1, 2, 3, 4, 12, 22, 23)] == [0, 1, 1, 0, 1, 1, 1, 0]
    assert [L.sws_test_transfer(i, 0) for i in (0, 1, 2, 3, 8, 9, 10, 11, 16, 18, 19)] == [0, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0]
    a = S.apply_props(S.SwsFrameView(), {}); b = S.apply_props(S.SwsFrameView(), {})   # av_frame_alloc() defaults: *_UNSPECIFIED
    z = S.SwsFrameView()
    z.width, z.height, z.format = 64, 32, P["yuv420p"]
    assert not L.sws_test_frame(C.byref(z), 0)          # all-zero colour fields: AVCOL_PRI_RESERVED0 is refused (format.c:648-653)
    a.width, a.height, a.format = 64, 32, P["yuv420p"]
    b.width, b.height, b.format = 64, 32, P["yuv420p"]
    assert L.sws_test_frame(C.byref(a), 0) and L.sws_is_noop(C.byref(a), C.byref(b))
    b.format = P["nv12"]
    assert not L.sws_is_noop(C.byref(a), C.byref(b))
    b.format = 11
    assert not L.sws_test_frame(C.byref(b), 1)
    assert L.sws_receive_slice_alignment(None) == 1


FILTER_CASES = [
    ("yuv420p", 64, 48, "yuv420p", 96, 80, LA.SWS_BICUBIC, (1.2, 0.8, 0.0, 0.0, 0.0, 0.0)),
    ("yuv420p", 96, 64, "rgb24", 64, 40, LA.SWS_BILINEAR, (0.0, 0.0, 0.6, 0.3, 1.0, 1.0)),
    ("yuv444p", 80, 60, "yuv420p", 80, 60, LA.SWS_LANCZOS, (2.0, 2.0, 0.5, 0.0, 0.0, 0.0)),
    ("rgb24", 72, 50, "yuv420p", 100, 30, LA.SWS_BICUBIC | LA.SWS_ACCURATE_RND, (0.7, 1.5, 0.0, 0.4, 2.0, 0.0)),
]


@pytest.mark.parametrize("case", FILTER_CASES, ids=lambda c: f"{c[0]}-{c[3]}-{c[2]}to{c[5]}")
def test_filter_banks_with_srcfilter_match_oracle(hiplib, case):
    """initFilter's SwsFilter convolution (utils.c:820-870): product host init == oracle, coefficient for coefficient."""
    sf, sw, sh, df, dw, dh, flags, fp = case
    L = hiplib
    f = L.sws_getDefaultFilter(*fp, 0)
    o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, src_filter=S.filter_to_dict(f))
    ctx = LA.SwsContext(sw, sh, sf, dw, dh, df, flags, src_filter=f, threads=1)
    L.sws_freeFilter(f)       # the context keeps its own copy of the vectors (utils.c:1290-1320 consumes them during init)
    for which in range(4):
        a, b = o.filter(which), ctx.filter(which)
        assert a[0] == b[0] and a[0] > 0
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # different banks than without the filter, and identity vectors change nothing
    plain = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    one = {"lumH": [1.0], "lumV": [1.0], "chrH": [1.0], "chrV": [1.0]}
    ident = OL.Oracle(sw, sh, sf, dw, dh, df, flags, src_filter=one)
    assert o.path() == "main" and any(o.filter(w)[0] != plain.filter(w)[0] or not np.array_equal(o.filter(w)[1], plain.filter(w)[1]) or
                                   not np.array_equal(o.filter(w)[2], plain.filter(w)[2]) for w in range(4))
    for w in range(4):
        assert np.array_equal(plain.filter(w)[1], ident.filter(w)[1]) and np.array_equal(plain.filter(w)[2], ident.filter(w)[2])
    ctx.close()
