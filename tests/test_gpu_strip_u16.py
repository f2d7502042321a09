"""-m gpu parity for sources with samples of 16 significant bits behind the marching strip kernels (round 5).

v_dot2_i32_i16 multiplies signed 16-bit operands, so rounds 1 - 4 kept yuv4xxp16 / gray16 / p016 and the 16-bit RGB families (rgb48 / rgba64, gbrp16,
gbrpf32) on the tile and element-per-thread kernels.  The register-staged strip kernels now flip the samples' top bit while staging (s' = s - 32768) and
start every horizontal chain from 32768 * (the column's tap sum), which is the same 32-bit sum (strip_hstage_b, kernels_strip.hpp): hScale16To15_c /
hScale16To19_c (swscale.c:69-125).  The RGB families come through the per-kind reader pre-pass (sws_k_read16_kind), whose lines have 16 significant bits
for them.  Every case is compared with the oracle; `no_strip_u16 = 1` gives the routes of rounds 1 - 4."""
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS, SWS_SPLINE
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
T0 = {"strip_min_w": 0}
OLD = dict(T0, no_strip_u16=1)

SRC_YUV = ["yuv420p16le", "yuv422p16le", "yuv444p16le", "p016le", "p216le", "p416le", "gray16le", "yuv420p16be"]
SRC_RGB = ["rgb48le", "bgr48le", "rgba64le", "bgra64le", "gbrp16le", "gbrpf32le", "gbrapf32le", "rgb48be"]
DST = ["yuv420p", "yuv422p", "yuv444p", "nv12", "yuv420p10le", "p010le", "yuv444p12le", "yuv420p16le", "p016le", "gray8", "gray10le", "yuyv422", "rgb565le", "ayuv"]
GEOMS = [(644, 70, 324, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (640, 48, 640, 24, SWS_BICUBIC), (640, 48, 320, 48, SWS_LANCZOS),
         (640, 48, 640, 48, SWS_BICUBIC)]


@pytest.mark.parametrize("form", ["u16", "old"])
@pytest.mark.parametrize("sfmt", SRC_YUV)
def test_planar_and_semi_planar_16_bit_sources(form, sfmt):
    for dfmt in DST:
        if ("gray" in sfmt) != ("gray" in dfmt):
            continue
        for (sw, sh, dw, dh, fl) in GEOMS:
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0 if form == "u16" else OLD)
            if form == "u16" and (sw != dw or sh != dh) and dfmt in ("yuv420p", "nv12", "yuv420p10le", "p010le", "yuv420p16le", "p016le", "gray8", "gray10le") and fl != SWS_LANCZOS:
                assert "strip" in path, (sfmt, dfmt, sw, dw, path)


@pytest.mark.parametrize("form", ["u16", "old"])
@pytest.mark.parametrize("sfmt", SRC_RGB)
def test_16_bit_rgb_sources(form, sfmt):
    for dfmt in ("yuv420p", "nv12", "yuv444p", "yuv420p10le", "p010le", "yuv422p", "yuvj420p"):
        for (sw, sh, dw, dh, fl) in GEOMS:
            path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0 if form == "u16" else OLD)
            if form == "u16" and (sw != dw) and sfmt not in ("rgba64le", "bgra64le", "gbrapf32le") and fl == SWS_BICUBIC:
                assert "rgbread" in path, (sfmt, dfmt, sw, dw, path)


@pytest.mark.parametrize("fl", [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND])
def test_scalers_and_geometries(fl):
    for (sw, sh, dw, dh) in ((1280, 72, 640, 36), (640, 36, 1280, 72), (900, 40, 452, 33), (770, 33, 384, 47), (1920, 30, 1280, 20), (700, 64, 700, 32), (1280, 96, 320, 24), (2560, 64, 320, 8)):
        for sfmt, dfmt in (("yuv420p16le", "yuv420p"), ("p016le", "nv12"), ("yuv444p16le", "yuv420p10le"), ("rgb48le", "yuv420p"), ("gbrpf32le", "nv12"), ("gray16le", "gray8"),
                           ("yuv420p16le", "yuv420p16le"), ("yuv422p16le", "p016le")):
            run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw, tune=T0)


def test_extreme_samples_and_options():
    """all-ones / all-zero pictures (the bias arithmetic at its ends), range conversion, shifted chroma positions"""
    import numpy as np
    import oracle_lib as OL
    for fill in (0xFF, 0x00, 0x80, 0x7F):
        for sfmt, dfmt in (("yuv420p16le", "yuv420p"), ("yuv444p16le", "yuv444p16le"), ("p016le", "p010le")):
            _orig = OL.fill_random
            try:
                def const_fill(frame, seed, _f=fill):
                    for pl in frame.planes:
                        pl[:] = _f
                    return frame
                OL.fill_random = const_fill
                run_case(640, 48, sfmt, 320, 24, dfmt, SWS_LANCZOS | BX, tune=T0)
                run_case(640, 48, sfmt, 960, 72, dfmt, SWS_SPLINE | BX, tune=T0)
            finally:
                OL.fill_random = _orig
    for (sr, dr) in ((0, 1), (1, 0)):
        opts = dict(dither=1, src_range=sr, dst_range=dr, src_h_chr_pos=0, src_v_chr_pos=128, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
        run_case(1280, 72, "yuv420p16le", 640, 36, "yuv420p", SWS_BICUBIC | BX, seed=3 + sr, opts=opts, tune=T0)
        run_case(1280, 72, "rgb48le", 640, 36, "yuv420p", SWS_BICUBIC | BX, seed=5 + sr, opts=opts, tune=T0)


def test_full_size_frames():
    assert "strip" in run_case(3840, 2160, "yuv420p16le", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=2)[0]
    assert "strip" in run_case(3840, 2160, "p016le", 1920, 1080, "nv12", SWS_BILINEAR | BX, seed=3, device_frames=False)[0]
    assert "rgbread" in run_case(1920, 1080, "rgb48le", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=4)[0]
    assert "rgbread" in run_case(1920, 1080, "gbrpf32le", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=5)[0]
    assert "strip" in run_case(1920, 1080, "gray16le", 960, 540, "gray8", SWS_BICUBIC | BX, seed=6)[0]


def test_16_bit_sources_into_packed_rgb():
    """yuv4xxp16 / p016 into 24 / 32 bpp RGB through the LUT writers (strip kernels' sums + sws_k_lut_rgb) and, with full chroma, sws_k_fullchr_rgb; rgb48 into
    bgra / rgb24 (RGB -> RGB: forced full chroma)"""
    for sfmt in ("yuv420p16le", "yuv422p16le", "p016le", "yuv444p16le", "rgb48le", "gbrp16le"):
        for dfmt in ("rgb24", "bgra", "bgr24", "argb", "gbrp"):
            for (sw, sh, dw, dh, fl) in ((644, 70, 324, 36, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (640, 48, 640, 48, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0)
                if (sw, dw) == (400, 332) and sfmt != "gbrp16le":
                    assert "strip" in path, (sfmt, dfmt, path)
    assert run_case(3840, 2160, "p016le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=9)[0].endswith("+lut_rgb")
