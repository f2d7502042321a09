"""-m gpu parity for the marching strip kernel (kernels_strip.hpp) at oracle-friendly sizes.  The library keeps narrow
pictures on the tile kernel (option "strip_min_w", default 320 output columns); these tests lower the threshold so that the
strip kernel itself is compared with the oracle: scalers, ratios, ragged widths, every planar / semi-planar writer."""
import pytest

from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA,
                           SWS_GAUSS, SWS_SPLINE)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT


STRIP = {"strip_min_w": 0}


SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "yuv420p10le", "yuv422p12le", "yuv444p9le", "yuv420p14le"]
DST = ["yuv420p", "yuv444p", "yuv411p", "yuv420p10le", "yuv444p12le", "nv12", "nv21", "nv16", "p010le", "p012le", "p210le"]


@pytest.mark.parametrize("sfmt", SRC)
@pytest.mark.parametrize("dfmt", DST)
def test_strip_formats(sfmt, dfmt):
    path, _ = run_case(322, 130, sfmt, 200, 74, dfmt, SWS_BICUBIC | BX, seed=5, tune=STRIP)
    if not (dfmt == "yuv411p" and "444" in sfmt):      # 4:4:4 -> 4:1:1 chroma needs more than 16 horizontal taps: tile kernel
        assert path == "main:strip_march"


@pytest.mark.parametrize("flags", [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE],
                         ids=["point", "area", "bilinear", "bicubic", "gauss", "lanczos", "spline"])
@pytest.mark.parametrize("geom", [(640, 96, 320, 48), (320, 48, 640, 96), (700, 50, 131, 77), (131, 77, 700, 50), (1030, 40, 258, 20),
                                  (300, 200, 300, 77), (257, 33, 513, 33), (2100, 24, 520, 12)],
                         ids=lambda g: f"{g[0]}x{g[1]}to{g[2]}x{g[3]}")
def test_strip_scalers_and_geometries(flags, geom):
    sw, sh, dw, dh = geom
    for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv420p10le", "p010le")):
        path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, flags | BX, seed=11, tune=STRIP)
        assert path in ("main:strip_march", "main:fused_tile_dot2", "main:fused_tile", "main:two_pass", "main:fused_generic_unity"), path


def test_strip_is_used_for_wide_pictures_by_default():
    path, _ = run_case(2048, 24, "yuv420p10le", 1024, 12, "p010le", SWS_LANCZOS | BX, seed=2)
    assert path == "main:strip_march"
    path, _ = run_case(640, 48, "yuv420p", 320, 24, "yuv420p", SWS_BILINEAR | BX, seed=2)      # (from 320 columns on: tools/narrow_shapes_times.py)
    assert path == "main:strip_march"
    path, _ = run_case(480, 48, "yuv420p", 240, 24, "yuv420p", SWS_BILINEAR | BX, seed=2)
    assert path != "main:strip_march"


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["nv12", "nv21", "p010le", "p012le", "p010be", "nv16", "p210le", "nv24", "p410le"])
@pytest.mark.parametrize("dst", ["yuv420p", "nv12", "nv21", "yuv420p10le", "p010le", "yuv422p", "yuv444p"])
def test_semi_planar_sources_take_the_strip_kernel(src, dst):
    """nv12 / nv21 / p010 / p012 sources and their 4:2:2 / 4:4:4 twins: the strip kernel de-interleaves plane 1 while staging (nvXXtoUV_c,
    input.c:926-948) and shifts the p01x samples down (p010LEToY_c / p010LEToUV_c, :950-1008)."""
    for (sw, sh, dw, dh, fl) in ((1280, 96, 1024, 64, SWS_BICUBIC), (1300, 70, 1030, 46, SWS_BILINEAR), (1024, 64, 2050, 130, SWS_LANCZOS),
                                 (1922, 50, 1280, 34, SWS_BICUBIC | SWS_ACCURATE_RND), (3840, 40, 1920, 20, SWS_AREA)):
        path, _ = run_case(sw, sh, src, dw, dh, dst, fl | BX, seed=sw)
        # the same kernel a planar 4:2:0 source of this geometry gets (a one-tap vertical chroma filter, e.g., keeps the tile kernel)
        twin = {"nv12": "yuv420p", "nv21": "yuv420p", "p010le": "yuv420p10le", "p012le": "yuv420p12le", "p010be": "yuv420p10le", "nv16": "yuv422p",
                "p210le": "yuv422p10le", "nv24": "yuv444p", "p410le": "yuv444p10le"}[src]
        tpath = run_case(sw, sh, twin, dw, dh, dst, fl | BX, seed=sw)[0]
        # (where the strip plan does not fit -- a 4:4:4 chroma window of more than 64 chunks -- the planar twin falls back to the dot2 tile
        #  kernel, which does not de-interleave: the semi-planar source takes the plain tile kernel)
        assert path == tpath or (tpath == "main:fused_tile_dot2" and path == "main:fused_tile"), (path, tpath, sw, dw)
    if src not in ("nv24", "p410le") or dst == "yuv444p":     # 4:4:4 -> 4:2:0 / 4:2:2 bicubic chroma needs more than 16 horizontal taps
        assert run_case(640, 96, src, 320, 64, dst, SWS_BICUBIC | BX, seed=3, tune=dict(strip_min_w=64))[0] == "main:strip_march"


@pytest.mark.parametrize("flags", [SWS_BICUBIC, SWS_BILINEAR, SWS_AREA, SWS_GAUSS], ids=["bicubic", "bilinear", "area", "gauss"])
def test_long_vertical_chroma_filters(flags):
    """a 4:1 vertical chroma step (packed RGB or 4:2:2 sources into a 4:2:0 picture of half the size): 17 .. 24 vertical chroma taps, the strip
    kernel's chroma instantiations with a ring of 12 row pairs (register-staged and LDS-DMA forms), both strip widths"""
    for tune in (STRIP, dict(strip_min_w=0, strip_cols_c=1)):
        for sfmt, dfmt in (("yuv422p", "yuv420p"), ("yuv422p10le", "yuv420p"), ("yuv422p", "nv12"), ("yuv422p12le", "p010le"), ("nv16", "yuv420p"), ("rgb24", "yuv420p"), ("bgra", "nv12")):
            for (sw, sh, dw, dh) in ((640, 128, 320, 64), (1288, 96, 644, 48), (520, 200, 300, 100)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, flags | BX, seed=sw + dh, tune=tune)
                if flags == SWS_BICUBIC and (sw, dw) != (520, 300):
                    assert path in ("main:strip_march", "main:rgbread+strip_march", "main:strip_rgbsrc"), (path, sfmt, dfmt, sw, dw)
    # full size: 4K capture into a 1080p 4:2:0 picture
    assert run_case(3840, 2160, "bgra", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=5)[0] == "main:strip_rgbsrc"
    assert run_case(3840, 2160, "yuv422p10le", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=6)[0] == "main:strip_march"


@pytest.mark.parametrize("sfmt", ["gray8", "gray10le", "gray12le", "gray14le"])
@pytest.mark.parametrize("dfmt", ["gray8", "gray9le", "gray10le", "gray12le"])
def test_gray_to_gray_takes_the_luma_launch(sfmt, dfmt):
    """one plane: the strip kernel's luma launch alone (gray -> gray, 8 .. 14 bit; gray16 keeps the 19-bit generic path)"""
    for (sw, sh, dw, dh, fl) in ((640, 96, 320, 48, SWS_BICUBIC), (322, 130, 200, 74, SWS_LANCZOS), (320, 48, 640, 96, SWS_BILINEAR), (1922, 50, 1280, 34, SWS_BICUBIC | SWS_ACCURATE_RND)):
        path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw, tune=STRIP)
        assert path == "main:strip_march", (path, sfmt, dfmt, sw, dw)
    assert run_case(1920, 1080, sfmt, 1280, 720, dfmt, SWS_BICUBIC | BX, seed=9)[0] == "main:strip_march"
    run_case(640, 96, sfmt, 320, 48, "gray16le", SWS_BICUBIC | BX, seed=3, tune=STRIP)
    run_case(640, 96, "gray16le", 320, 48, dfmt, SWS_BICUBIC | BX, seed=3, tune=STRIP)


def test_one_tap_vertical_filters():
    """a plane whose vertical filter is the identity next to scaled ones: yuv2plane1's (s + d) >> 7 is the X form with the one tap 4096
    (4:2:0 -> 4:2:2 at half the height: chroma rows unscaled; horizontal-only scaling: both planes)"""
    for sfmt, dfmt in (("yuv420p", "yuv422p"), ("yuv420p10le", "yuv422p10le"), ("yuv420p10le", "yuv422p"), ("nv12", "nv16"), ("yuv420p", "yuv422p12le")):
        for (sw, sh, dw, dh, fl) in ((640, 96, 320, 48, SWS_BICUBIC), (1288, 96, 644, 48, SWS_LANCZOS), (640, 96, 400, 48, SWS_BILINEAR)):
            assert run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw, tune=STRIP)[0] == "main:strip_march", (sfmt, dfmt, sw, dw)
    for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv422p10le", "yuv422p"), ("yuv444p", "yuv444p12le"), ("nv12", "yuv420p"), ("gray8", "gray8")):
        for (sw, sh, dw, dh, fl) in ((1440, 64, 1920, 64, SWS_BICUBIC), (640, 50, 320, 50, SWS_BICUBIC), (322, 33, 200, 33, SWS_LANCZOS)):
            assert run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw, tune=STRIP)[0] == "main:strip_march", (sfmt, dfmt, sw, dw)
    assert run_case(3840, 2160, "yuv420p", 1920, 1080, "uyvy422", SWS_BICUBIC | BX, seed=5)[0] == "main:strip_march+join422"
    assert run_case(1440, 1080, "yuv420p", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=6)[0] == "main:strip_march"


def test_one_tap_rows_with_the_tap_4095():
    """initFilter's error-diffused normalisation leaves 4095 in some rows of a one-tap vertical bank (tiny source heights with a shifted chroma
    position): yuv2plane1 never looks at the coefficient, the semi-planar chroma writers and the packed X forms do"""
    opts = dict(dither=2, src_range=0, dst_range=0, src_h_chr_pos=0, src_v_chr_pos=256, dst_h_chr_pos=0, dst_v_chr_pos=-513)
    for sfmt, dfmt in (("yuyv422", "yuyv422"), ("yuv422p", "yuv422p"), ("yuv422p", "nv16"), ("yuv422p", "uyvy422"), ("yuv444p12le", "yuv420p10be"), ("yuv422p", "p210le"),
                       ("yuv444p", "rgb24"), ("yuv422p", "bgra")):
        for (sw, sh, dw, dh) in ((60, 3, 302, 44), (340, 2, 352, 25), (64, 3, 128, 3), (128, 2, 64, 9)):
            run_case(sw, sh, sfmt, dw, dh, dfmt, SWS_BICUBIC | SWS_ACCURATE_RND | BX, seed=sw, opts=opts, tune=STRIP)
            run_case(sw, sh, sfmt, dw, dh, dfmt, SWS_LANCZOS | BX, seed=dw, opts=dict(opts, src_v_chr_pos=128), tune=STRIP)


@pytest.mark.parametrize("flags", [SWS_BICUBIC, SWS_BILINEAR, SWS_AREA, SWS_GAUSS, SWS_LANCZOS, SWS_SPLINE, SWS_BICUBIC | SWS_ACCURATE_RND],
                         ids=["bicubic", "bilinear", "area", "gauss", "lanczos", "spline", "bicubic_ar"])
@pytest.mark.parametrize("geom", [(1024, 256, 256, 64), (1280, 360, 256, 60), (1536, 384, 256, 64), (1792, 224, 256, 32), (1024, 128, 341, 43), (1000, 250, 203, 51),
                                  (1024, 64, 256, 64), (512, 256, 512, 64), (1920, 270, 426, 60), (1100, 140, 157, 20)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_long_filters(flags, geom):
    """ratios of 3:1 .. 7:1 (the lower rungs of an ABR ladder, thumbnails): filters of 17 .. 32 taps take the strip kernel's long form (16 tap pairs each
    way, a ring of 16 row pairs, strips of 128 / 64 columns); longer ones still fall back"""
    sw, sh, dw, dh = geom
    for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv420p10le", "nv12"), ("nv12", "yuv420p10le"), ("yuv444p", "yuv420p"), ("rgb24", "yuv420p"), ("bgra", "rgb24"),
                       ("yuv422p", "p010le"), ("yuv420p", "gbrp"), ("yuyv422", "yuv422p"), ("yuva420p", "yuva420p"),
                       ("yuv420p", "rgb24"), ("nv12", "bgra"), ("yuv420p10le", "rgba"), ("yuv422p", "bgr24"), ("p010le", "argb"), ("uyvy422", "rgb24"), ("yuvj420p", "bgr0")):
        run_case(sw, sh, sfmt, dw, dh, dfmt, flags | BX, seed=sw + dh, tune=STRIP)


def test_long_filters_take_the_strip_kernel():
    from librempeg_amd import SwsContext
    for (sw, sh, sf, dw, dh, df, fl) in ((3840, 2160, "yuv420p", 960, 540, "yuv420p", SWS_BICUBIC), (3840, 2160, "yuv420p", 640, 360, "yuv420p", SWS_BICUBIC),
                                         (3840, 2160, "yuv420p", 1280, 720, "nv12", SWS_LANCZOS), (1920, 1080, "yuv420p10le", 426, 240, "yuv420p", SWS_BICUBIC),
                                         (3840, 2160, "rgb24", 960, 540, "yuv420p", SWS_BICUBIC), (3840, 2160, "bgra", 960, 540, "bgra", SWS_BICUBIC)):
        c = SwsContext(sw, sh, sf, dw, dh, df, fl | BX)      # (the planner's own thresholds: 64 columns for the long forms)
        assert "strip_march" in c.path() and c.kernel_name() == "sws_k_strip_long", (c.path(), c.kernel_name(), sf, df, dw)
        c.close()
    path, _ = run_case(3840, 2160, "yuv420p", 960, 540, "yuv420p", SWS_BICUBIC | BX, seed=21)
    assert path == "main:strip_march"
    path, _ = run_case(3840, 2160, "yuv420p10le", 640, 360, "p010le", SWS_BICUBIC | BX, seed=22)
    assert path == "main:strip_march"
    path, _ = run_case(3840, 2160, "rgb24", 960, 540, "yuv420p", SWS_BICUBIC | BX, seed=23)
    assert path == "main:rgbread+strip_march"
    path, _ = run_case(3840, 2160, "yuv420p", 960, 540, "rgb24", SWS_BICUBIC | BX, seed=25)      # the LUT writers: raw sums + sws_k_lut_rgb
    assert path == "main:strip_march+lut_rgb"
    path, _ = run_case(1920, 1080, "nv12", 320, 180, "bgra", SWS_BICUBIC | BX, seed=26)
    assert path == "main:splitnv+strip_march+lut_rgb" or path == "main:strip_march+lut_rgb", path
    path, _ = run_case(3840, 2160, "bgra", 640, 360, "bgra", SWS_BICUBIC | BX, seed=24)
    assert path == "main:rgbread+strip_march+fullchr_rgb"


UNITY_PAIRS = [("p010le", "yuv420p10le"), ("p010le", "yuv420p"), ("nv12", "yuv420p10le"), ("yuv420p10le", "nv12"), ("nv12", "nv21"), ("nv21", "yuv420p10le"), ("p010le", "nv12"),
               ("yuv420p", "yuv420p12le"), ("yuv422p10le", "nv16"), ("bgra", "yuv444p10le"), ("rgb24", "yuv422p10le"), ("nv12", "yuva420p"), ("p012le", "yuv420p10le"),
               ("yuv444p", "yuv444p10le"), ("yuv420p10le", "p012le"), ("nv16", "yuv422p10le"), ("p210le", "yuv422p"), ("gbrp", "yuv444p12le")]


@pytest.mark.parametrize("pair", UNITY_PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_unity_conversions_without_a_special_converter(pair):
    """same-size conversions whose four filters are the identity and for which the reference has no unscaled converter (a hardware decoder's p010 / nv12 into the
    planar layouts of the software encoders and back): per-sample work of the scaler chain, one-tap banks on the strip kernels; and vertical-only scaling"""
    sf, df = pair
    for (sw, sh, dw, dh, fl) in ((256, 64, 256, 64, SWS_BICUBIC), (322, 50, 322, 50, SWS_BILINEAR), (1920, 32, 1920, 32, SWS_BICUBIC), (256, 64, 256, 40, SWS_BICUBIC),
                                 (256, 40, 256, 96, SWS_LANCZOS), (1280, 36, 1280, 24, SWS_AREA)):
        path, _ = run_case(sw, sh, sf, dw, dh, df, fl | BX, seed=sw + dh, tune=STRIP if sw < 1024 else None)
        if (sw, sh) == (1920, 32):
            assert "strip_march" in path or "strip_chroma" in path or path == "main:strip_rgbsrc" or path.startswith("unscaled:"), (path, sf, df)   # (p010le -> nv12: planarCopy)


def test_p010_to_rgb_at_the_same_size():
    for df in ("bgra", "rgb24", "argb"):
        assert run_case(1920, 1080, "p010le", 1920, 1080, df, SWS_BICUBIC | BX, seed=31)[0] == "main:nvdirect+strip_rgb"
        assert run_case(322, 50, "p012le", 322, 50, df, SWS_BICUBIC | BX, seed=32, tune=STRIP)[0] == "main:nvdirect+strip_rgb"
    assert run_case(1920, 1080, "p010le", 1920, 1080, "yuv420p10le", SWS_BICUBIC | BX, seed=33)[0] == "main:strip_march"
    assert run_case(1920, 1080, "nv12", 1920, 1080, "yuv420p10le", SWS_BICUBIC | BX, seed=34)[0] == "main:strip_march"
    assert run_case(1920, 1080, "nv12", 1920, 1080, "bgr0", SWS_BICUBIC | BX, seed=35)[0] == "main:fused_rgb_unity"      # C4 keeps its kernel


@pytest.mark.parametrize("flags", [SWS_BICUBIC, SWS_BILINEAR, SWS_AREA, SWS_GAUSS, SWS_LANCZOS, SWS_BICUBIC | SWS_ACCURATE_RND],
                         ids=["bicubic", "bilinear", "area", "gauss", "lanczos", "bicubic_ar"])
@pytest.mark.parametrize("geom", [(2048, 288, 256, 36), (2560, 360, 256, 36), (3072, 240, 256, 20), (3840, 270, 256, 18), (2048, 128, 300, 19), (2000, 250, 203, 21),
                                  (2048, 32, 256, 32), (512, 400, 512, 40), (1920, 540, 240, 68), (2200, 140, 257, 20)],
                         ids=lambda g: f"{g[0]}x{g[1]}-{g[2]}x{g[3]}")
def test_extra_long_filters(flags, geom):
    """ratios of 8:1 .. 15:1 (thumbnails, preview sprites): filters of 33 .. 62 taps take the extra-long form (32 tap pairs each way, strips of 64 columns);
    still longer ones fall back"""
    sw, sh, dw, dh = geom
    for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv420p10le", "nv12"), ("nv12", "yuv420p10le"), ("yuv444p", "yuv420p"), ("rgb24", "yuv444p"), ("bgra", "rgb24"),
                       ("yuv420p", "gbrp"), ("yuv420p", "rgb24"), ("nv12", "bgra"), ("yuva420p", "yuva420p")):
        run_case(sw, sh, sfmt, dw, dh, dfmt, flags | BX, seed=sw + dh, tune=STRIP)


def test_extra_long_filters_take_the_strip_kernel():
    from librempeg_amd import SwsContext
    for (sw, sh, sf, dw, dh, df, fl) in ((3840, 2160, "yuv420p", 480, 270, "yuv420p", SWS_BICUBIC), (3840, 2160, "yuv420p", 320, 180, "yuv420p", SWS_BICUBIC),
                                         (1920, 1080, "nv12", 256, 144, "nv12", SWS_LANCZOS), (3840, 2160, "yuv420p", 480, 270, "rgb24", SWS_BICUBIC)):
        c = SwsContext(sw, sh, sf, dw, dh, df, fl | BX)
        assert "strip_march" in c.path() and c.kernel_name() == "sws_k_strip_xlong", (c.path(), c.kernel_name(), sf, df, dw)
        c.close()
    assert run_case(3840, 2160, "yuv420p", 480, 270, "yuv420p", SWS_BICUBIC | BX, seed=61)[0] == "main:strip_march"
    assert run_case(3840, 2160, "yuv420p", 320, 180, "rgb24", SWS_BICUBIC | BX, seed=62)[0] == "main:strip_march+lut_rgb"
    assert run_case(1920, 1080, "yuv420p10le", 256, 144, "yuv420p", SWS_LANCZOS | BX, seed=63)[0] == "main:strip_march"


def test_small_thumbnails_take_the_long_forms():
    """outputs of 64 .. 319 columns with filters of more than 16 taps (160 x 90 from 1080p): the long forms' own width threshold"""
    assert run_case(1920, 1080, "yuv420p", 160, 90, "yuv420p", SWS_BICUBIC | BX, seed=71)[0] == "main:strip_march"
    assert run_case(1920, 1080, "yuv420p", 128, 72, "rgb24", SWS_BICUBIC | BX, seed=72)[0] == "main:strip_march+lut_rgb"
    assert run_case(1280, 720, "nv12", 160, 90, "yuv420p", SWS_LANCZOS | BX, seed=73)[0] == "main:strip_march"
    assert run_case(1280, 720, "bgra", 160, 90, "bgra", SWS_BICUBIC | BX, seed=74)[0] == "main:rgbread+strip_march+fullchr_rgb"
    for (sw, sh, dw, dh) in ((640, 360, 64, 36), (650, 365, 70, 40), (1000, 300, 100, 30), (520, 130, 65, 26), (4 * 163, 90, 66, 9)):
        for sf, df in (("yuv420p", "yuv420p"), ("yuv420p10le", "nv12"), ("rgb24", "yuv420p"), ("yuv420p", "bgra"), ("yuv420p", "rgb24"), ("yuva420p", "yuva420p"), ("bgra", "gbrp")):
            run_case(sw, sh, sf, dw, dh, df, SWS_BICUBIC | BX, seed=sw + dh)
    assert run_case(600, 300, "yuv420p", 60, 30, "yuv420p", SWS_BICUBIC | BX, seed=75)[0] != "main:strip_march"      # below 64 columns
