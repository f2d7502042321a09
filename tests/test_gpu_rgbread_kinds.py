"""-m gpu parity for the reader pre-pass of the RGB sources beyond the 8-bit ones (dev_prepare_on: rgbread_kindN): x2rgb10 / x2bgr10, the 16 / 15 / 12 bpp
formats (rgb16_32ToY/UV(_half)_c_template, input.c:264-412) and planar RGB of 9 - 14 bits (planar_rgb16_s16_to_y / _uv, :1216-1270) deliver the same 15-bit
lines to hScale16To15_c (sh = 13) as the 8-bit RGB readers; the per-kind element-per-thread reader (k_generic_kinds.hip sws_k_read16_kind) writes them as
planes of a working picture and the strip kernels scale those like a planar 16-bit source (main:rgbread+strip_march[+...]).  Every case also runs with the
option no_rgbread_kinds = 1 (the tile / two-pass kernels these sources had before)."""
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_AREA, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_FULL_CHR_H_INP
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
TUNE = dict(strip_min_w=0)
OLD = dict(strip_min_w=0, no_rgbread_kinds=1)

SRC = ["x2rgb10le", "x2bgr10le", "rgb565le", "bgr565le", "rgb555le", "bgr555le", "rgb444le", "bgr444le", "rgb565be", "gbrp9le", "gbrp10le", "gbrp12le", "gbrp14le", "gbrp10be",
       "gbrp10msble", "gbrp12msble",
       "y210le", "y212le", "xv30le", "v30xle", "xv36le", "xv36be",
       "vyu444", "vuyx", "ayuv", "vuya", "uyva", "gbrap10le", "gbrap12le"]   # (packed YUV of 10 / 12 bits: the lines of a planar yuv422p10 / yuv444p10 / ...12 picture)
DST = ["yuv420p", "yuv422p", "yuv444p", "nv12", "yuv420p10le", "p010le", "bgra", "rgb24", "gbrp", "uyvy422"]


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_formats(src, dst):
    for (sw, sh, dw, dh) in ((256, 64, 192, 48), (320, 50, 512, 80), (132, 33, 66, 17)):
        path, _ = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=TUNE)
        if dst in DST[:6]:      # (the YUV destinations; the RGB / packed 4:2:2 ones add their own planner conditions behind the strip kernels)
            assert "rgbread" in path or (src in SRC[:2] and path == "main:strip_rgbsrc"), (path, sw, sh, dw, dh)   # (x2rgb10 / x2bgr10 into half-width chroma: the one-launch kernel reads them itself)
            if src in SRC[:2]:
                assert "rgbread" in run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=dict(TUNE, no_strip_rgbsrc=1))[0]
        old, _ = run_case(sw, sh, src, dw, dh, dst, SWS_BICUBIC | BX, seed=sw, tune=OLD)
        assert "rgbread" not in old, old


@pytest.mark.parametrize("flags", [SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_LANCZOS, SWS_BICUBIC | SWS_ACCURATE_RND, SWS_BILINEAR | SWS_FULL_CHR_H_INP],
                         ids=["area", "bilinear", "bicubic", "lanczos", "accurate", "fullinp"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (640, 96, 426, 64), (260, 200, 520, 300), (1924, 34, 1282, 22), (64, 40, 1030, 44), (2052, 20, 1026, 10), (640, 48, 640, 48)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for src, dst in (("x2rgb10le", "yuv420p"), ("rgb565le", "nv12"), ("gbrp10le", "yuv422p10le"), ("gbrp12le", "bgra"), ("y210le", "yuv420p"), ("xv30le", "yuv420p10le"), ("xv36le", "nv12"), ("vuya", "yuv420p"), ("vyu444", "nv12")):
        run_case(sw, sh, src, dw, dh, dst, flags | BX, seed=7, tune=TUNE)


def test_planner_and_fallbacks():
    assert "rgbread" in run_case(1920, 54, "gbrp10le", 1280, 36, "yuv420p", SWS_BICUBIC | BX)[0]                   # wide enough without the option
    assert "rgbread" in run_case(642, 48, "rgb565le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]          # a width of 4 k + 2 (round 5)
    assert "rgbread" not in run_case(643, 48, "rgb565le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]      # odd width
    assert "rgbread" in run_case(640, 48, "gbrap10le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]         # a source with an alpha plane nobody reads
    assert "rgbread" not in run_case(640, 48, "gbrap10le", 320, 24, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0]    # ... and one the destination wants scaled
    assert "rgbread" not in run_case(640, 48, "ayuv", 320, 24, "yuva420p", SWS_BICUBIC | BX, tune=TUNE)[0]
    assert "rgbread" in run_case(640, 48, "gbrp16le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]          # 16-bit samples: 16-bit lines (round 5: strip_hstage_b)
    assert "rgbread" not in run_case(640, 48, "gbrp16le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_u16=1))[0]
    assert run_case(640, 48, "x2rgb10le", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=TUNE)[0] == "main:strip_rgbsrc"   # a range conversion (round 5: in the strip kernels; x2rgb10 / x2bgr10 read by the one-launch kernel)
    assert "rgbread" in run_case(640, 48, "x2rgb10le", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_rgbsrc=1))[0]
    assert "rgbread" in run_case(640, 48, "rgb565le", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=TUNE)[0]
    assert "rgbread" not in run_case(640, 48, "rgb565le", 320, 24, "yuvj420p", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_range=1))[0]
    assert "rgbread" in run_case(640, 48, "x2rgb10le", 320, 24, "yuv420p16le", SWS_BICUBIC | BX, tune=TUNE)[0]      # 19-bit intermediates: round 5 (sws_k_strip_wide on the reader planes)
    assert "rgbread" not in run_case(640, 48, "x2rgb10le", 320, 24, "yuv420p16le", SWS_BICUBIC | BX, tune=dict(TUNE, no_strip_wide=1))[0]
    assert "rgbread" not in run_case(640, 48, "y216le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]        # 16-bit samples
    assert "rgbread" not in run_case(640, 48, "ayuv64le", 320, 24, "yuv420p", SWS_BICUBIC | BX, tune=TUNE)[0]      # 16-bit samples, alpha
    assert "rgbread" in run_case(1920, 54, "y210le", 1280, 36, "yuv420p", SWS_BICUBIC | BX)[0]
    assert "rgbread" in run_case(1920, 54, "xv30le", 1280, 36, "p010le", SWS_LANCZOS | BX)[0]


def test_full_size_frames_host_frames_and_batches():
    assert "rgbread" in run_case(1920, 1080, "gbrp10le", 1280, 720, "yuv420p", SWS_BICUBIC | BX, seed=2)[0]
    assert "rgbread" in run_case(1920, 1080, "x2rgb10le", 2560, 1440, "nv12", SWS_BILINEAR | BX, seed=3)[0]
    assert "rgbread" in run_case(2560, 1440, "rgb565le", 1920, 1080, "bgra", SWS_LANCZOS | BX, seed=4, device_frames=False)[0]
    from test_gpu_unaligned_frames import run_odd
    for src, dst in (("rgb565le", "yuv420p"), ("gbrp10le", "nv12"), ("x2rgb10le", "bgra")):
        for pad, shift, flip in ((4, 4, 0), (0, 0, 3), (12, 8, 1)):
            run_odd(640, 40, src, 426, 26, dst, SWS_BICUBIC | BX, pad, shift, flip, tune=TUNE)


def test_packed_yuv_sources_into_packed_rgb_through_the_lut_epilogue():
    """round 5: y210 / xv30 / xv36 / vyu444 scaled into 24 / 32 bpp RGB through the LUT writers -- reader pre-pass, strip kernels' sums, sws_k_lut_rgb"""
    for sfmt in ("y210le", "y212le", "xv30le", "xv36le", "vyu444", "vuyx"):
        for dfmt in ("bgra", "rgb24", "argb"):
            for (sw, sh, dw, dh, fl) in ((640, 48, 320, 24, SWS_BICUBIC), (644, 40, 516, 32, SWS_BILINEAR), (320, 24, 640, 48, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw + len(sfmt), tune=TUNE)
                if (sw, dw) == (640, 320) and sfmt in ("y210le", "y212le"):     # (4:4:4 sources into RGB: full chroma is forced, the other epilogue)
                    assert path.endswith("+lut_rgb"), (sfmt, dfmt, path)
