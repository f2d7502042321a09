"""The five classes in which round 2's restatements (oracle and product alike) differed from the real reference, as the round-2 review
measured them against a C-only libswscale; each rule is now restated from the reference's source in both, and compared here.

(a) alpha_blend on planar RGB + alpha sources: alphaless_fmt() has GBRAP / GBRAP10..16 rows (utils.c:1073-1085), ff_sws_alphablendaway
    handles planar RGB (alphablend.c:47-130, with "w = plane ? c->chrSrcW : src_w") and returns 0 (:176).
(b) gray8 -> byte RGB / gbrp / gbrap at the same size is palToRgbWrapper / palToGbrpWrapper with the grey ramp (swscale_unscaled.c:2619-2630,
    swscale.c:901-902): the sample is replicated whatever sws_setColorspaceDetails() said.
(c) SWS_SRC_V_CHR_DROP on planar RGB sources: chr_convert's plane-0 row comes from the start of its batch of lines (hscale.c:211-225).
(d) gamma_flag: gamma_convert rewrites source lines in place as the ring pulls them (gamma.c:31-58, slice.c:325-328); a line pulled again
    after a hole is converted twice (swscale.c:404-451).
(e) cascades run before the XYZ passes of scale_internal (swscale.c:1076-1082 vs :1126, :1194), and their children are created with the
    formats handle_formats() already aliased: an xyz12 picture on either side of a cascade is treated as rgb48."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import (SwsContext, HostFrame, DeviceFrame, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND,
                           SWS_POINT, SWS_FAST_BILINEAR, SWS_CS_ITU709, SWS_CS_ITU601, SWS_CS_BT2020)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT


def _ids(c):
    return f"{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}"


# ---- (a) ----
A_DIRECT = [("gbrap", "gbrp"), ("gbrap10le", "gbrp10le"), ("gbrap12le", "gbrp12le"), ("gbrap14le", "gbrp14le"), ("gbrap16le", "gbrp16le"),
            ("gbrap10be", "gbrp10le"), ("gbrap16be", "gbrp16le")]
A_CASCADE = [(112, 35, "gbrap16le", 112, 35, "bgr555be", SWS_BICUBIC), (49, 12, "gbrap", 49, 12, "gbrp", SWS_BICUBIC),
             (64, 40, "gbrap12be", 64, 40, "gbrp12be", SWS_BICUBIC), (97, 65, "gbrap", 64, 40, "yuv420p", SWS_BICUBIC),
             (96, 64, "gbrap10le", 128, 80, "rgb24", SWS_LANCZOS), (96, 64, "gbrap14le", 96, 64, "gbrp14le", SWS_FAST_BILINEAR)]


@pytest.mark.parametrize("mode", [1, 2], ids=["uniform", "checkerboard"])
@pytest.mark.parametrize("pair", A_DIRECT, ids=lambda p: f"{p[0]}-{p[1]}")
def test_gbrap_blendaway(pair, mode):
    sf, df = pair
    for (w, h, fl) in ((97, 67, SWS_BICUBIC), (64, 34, SWS_BICUBIC), (64, 34, SWS_FAST_BILINEAR), (49, 12, SWS_POINT)):
        path, opath = run_case(w, h, sf, w, h, df, fl | BX, seed=w + mode, opts=dict(alpha_blend=mode))
        assert (path, opath) == ("unscaled:alphablendaway", "alphablendaway")
        run_case(w, h, sf, w, h, df, fl | BX, seed=h, opts=dict(alpha_blend=mode), device_frames=False)


@pytest.mark.parametrize("mode", [1, 2], ids=["uniform", "checkerboard"])
@pytest.mark.parametrize("case", A_CASCADE, ids=_ids)
def test_gbrap_blend_cascade(case, mode):
    sw, sh, sf, dw, dh, df, flags = case
    path, opath = run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=13, opts=dict(alpha_blend=mode))
    if (sf, df) == ("gbrap", "gbrp") or (sf[:-2] == "gbrap14" and flags == SWS_FAST_BILINEAR):
        assert (path, opath) == ("unscaled:alphablendaway", "alphablendaway")
    else:
        assert (path, opath) == ("cascade", "cascade")


def test_blendaway_returns_zero_rows():
    """ff_sws_alphablendaway ends in `return 0` (alphablend.c:176) and scale_internal passes that on: sws_scale() reports 0 rows."""
    for sf, df in (("rgba", "rgb24"), ("gbrap", "gbrp"), ("yuva420p", "yuv420p")):
        o = OL.Oracle(64, 32, sf, 64, 32, df, SWS_BICUBIC | BX, alpha_blend=1)
        p = SwsContext(64, 32, sf, 64, 32, df, SWS_BICUBIC | BX, alpha_blend=1)
        src = OL.fill_random(OL.Frame(sf, 64, 32), 5)
        ref = OL.Frame(df, 64, 32)
        assert o.scale(src, ref) == 0
        hs = HostFrame(sf, 64, 32)
        for a, b in zip(hs.planes, src.planes):
            a[:] = b
        hd = HostFrame(df, 64, 32)
        assert p.scale(hs, hd) == 0
        assert all(np.array_equal(a[:, :rb], b[:, :rb]) for a, b, rb in zip(hd.planes, ref.planes, hd.row_bytes))


# ---- (b) ----
DETAILS = [None, (SWS_CS_ITU601, 0, SWS_CS_ITU601, 0, 4096, 1 << 16, 1 << 16), (SWS_CS_ITU709, 1, SWS_CS_BT2020, 0, -3000, 70000, 90000)]


@pytest.mark.parametrize("details", DETAILS, ids=["default", "brightness", "matrix"])
@pytest.mark.parametrize("df", ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "gbrp", "gbrap"])
def test_gray8_takes_the_palette_wrapper(df, details):
    for (w, h) in ((64, 32), (33, 17)):
        path, opath = run_case(w, h, "gray8", w, h, df, SWS_BICUBIC | BX, seed=w, colorspace=details)
        assert (path, opath) == ("unscaled:palToRgb", "palToRgb")
        run_case(w, h, "gray8", w, h, df, SWS_BICUBIC | BX, seed=h, colorspace=details, device_frames=False)


def test_gray8_matrix_cascade_keeps_luma():
    """gray8 137x89 -> yuvj420p with different matrices: context[0] of the YUV -> YUV cascade is gray8 -> bgr24 at the same size, i.e. the
    palette wrapper (luma untouched), not the scaler chain (the case the review found green on the GPU and wrong against the reference)."""
    run_case(137, 89, "gray8", 137, 89, "yuvj420p", SWS_BICUBIC | BX, seed=2, colorspace=(SWS_CS_BT2020, 0, SWS_CS_ITU601, 1, 0, 1 << 16, 1 << 16))
    run_case(64, 48, "gray8", 64, 48, "yuv444p", SWS_BILINEAR | BX, seed=3, colorspace=(SWS_CS_ITU709, 1, SWS_CS_ITU601, 0, 2000, 60000, 1 << 16))


# ---- (c) ----
C_CASES = [(45, 25, "gbrp", 45, 25, "ayuv64le", 0x60001), (27, 104, "gbrpf32le", 27, 104, "bgr24", 0x90004), (14, 68, "gbrp9le", 32, 28, "yuva422p10le", 0x20004),
           (96, 64, "gbrp", 64, 40, "yuv420p", 0x10000 | SWS_BICUBIC), (96, 64, "gbrp", 64, 40, "yuv420p", 0x30000 | SWS_BICUBIC),
           (96, 64, "gbrap", 128, 80, "yuva444p", 0x20000 | SWS_BILINEAR), (61, 77, "gbrp16le", 47, 29, "yuv422p16le", 0x20000 | SWS_POINT),
           (61, 77, "gbrp12be", 61, 77, "nv12", 0x10000 | SWS_FAST_BILINEAR), (50, 90, "gbrpf16le", 50, 30, "rgb24", 0x20000 | SWS_BICUBIC),
           (50, 90, "gbrapf32le", 70, 41, "bgra", 0x30000 | SWS_LANCZOS), (64, 200, "gbrp10le", 64, 37, "yuv420p10le", 0x10000 | SWS_POINT),
           (64, 200, "gbrp", 64, 51, "gray8", 0x20000 | SWS_FAST_BILINEAR)]


@pytest.mark.parametrize("case", C_CASES, ids=_ids)
def test_vchrdrop_planar_rgb(case):
    sw, sh, sf, dw, dh, df, flags = case
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=6)
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=7, device_frames=False)


# ---- (d) ----
D_CASES = [(157, 68, "yuva444p12le", 181, 31, "yuva422p", 0x80001), (124, 108, "y212le", 123, 95, "gray14le", 0xc0010),
           (96, 200, "rgb24", 64, 61, "rgb24", SWS_FAST_BILINEAR), (96, 200, "yuv420p", 128, 77, "rgba64le", SWS_POINT),
           (90, 131, "rgba64le", 45, 40, "bgra", SWS_FAST_BILINEAR), (90, 131, "gbrp10le", 90, 100, "yuv444p10le", SWS_POINT | SWS_ACCURATE_RND),
           (64, 97, "rgba", 64, 30, "rgba", SWS_BILINEAR), (200, 160, "rgb48be", 100, 37, "rgb48be", SWS_FAST_BILINEAR)]


@pytest.mark.parametrize("case", D_CASES, ids=_ids)
def test_gamma_cascade_with_holes(case):
    sw, sh, sf, dw, dh, df, flags = case
    path, opath = run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=8, opts=dict(gamma_flag=1))
    assert (path, opath) == ("cascade", "cascade")
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=9, opts=dict(gamma_flag=1), device_frames=False)


# ---- (e) ----
E_CASES = [(64, 48, "bayer_bggr8", 48, 36, "xyz12le", SWS_BICUBIC, {}), (64, 48, "bayer_rggb16le", 64, 48, "xyz12be", SWS_BICUBIC, {}),
           (64, 48, "bayer_gbrg16be", 80, 60, "xyz12le", SWS_BILINEAR, {}), (64, 48, "bayer_grbg8", 64, 48, "xyz12le", SWS_BICUBIC, {}),
           (96, 64, "xyz12le", 64, 40, "rgb24", SWS_BICUBIC, dict(gamma_flag=1)), (96, 64, "yuv420p", 64, 40, "xyz12be", SWS_BILINEAR, dict(gamma_flag=1)),
           (96, 64, "xyz12le", 64, 40, "xyz12le", SWS_BICUBIC, dict(gamma_flag=1)),
           (1040, 8, "xyz12le", 16, 8, "rgb24", SWS_BICUBIC, {}), (1040, 8, "yuv420p", 16, 8, "xyz12le", SWS_BICUBIC, {}),
           (96, 64, "rgba", 64, 40, "xyz12le", SWS_BICUBIC, dict(alpha_blend=1))]


@pytest.mark.parametrize("case", E_CASES, ids=lambda c: _ids(c) + ("-" + "-".join(c[7]) if c[7] else ""))
def test_xyz_around_cascades(case):
    sw, sh, sf, dw, dh, df, flags, opts = case
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=10, opts=opts)
    run_case(sw, sh, sf, dw, dh, df, flags | BX, seed=11, opts=opts, device_frames=False)
