"""-m gpu parity for MPEG <-> JPEG range conversion inside the marching strip kernels (round 5).

The reference converts the range on the h-scaled lines, between the horizontal and the vertical scaler (lum / chrRange{To,From}Jpeg_c,
swscale.c:163-209, called from hscale.c:61-63 and :195-197; constants from solve_range_convert, :577-660): int arithmetic on the int16 line,
the ToJpeg forms clip to 2^15 - 1, the alpha line is h-scaled by the luma function but not converted.  Rounds 1 - 4 kept every such context
(yuvj* on one side, src_range != dst_range, gray8 against limited-range YUV) on the tile and element-per-thread kernels; the strip kernels
now convert the packed {even row, odd row} dword of a column on its way into the register ring (strip_range, kernels_strip.hpp).
Every case is compared with the oracle, with the expected route asserted; `no_strip_range = 1` (the old routes) must give the same bytes."""
import numpy as np
import pytest

import oracle_lib as OL
from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_AREA, SWS_POINT, SWS_GAUSS
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT
T0 = {"strip_min_w": 0}
FORMS = {
    "planner": dict(T0),
    "short": dict(T0, no_strip_dma8=1),
    "general": dict(T0, no_strip_short=1),
    "old": dict(T0, no_strip_range=1),
}

# (source, destination): one side full range by its name (yuvj*, gray) or both by option below
PAIRS = [("yuvj420p", "yuv420p"), ("yuv420p", "yuvj420p"), ("yuvj422p", "yuv420p"), ("yuvj444p", "nv12"), ("yuvj420p", "yuv420p10le"), ("yuv420p10le", "yuvj420p"),
         ("yuv422p10le", "yuvj422p"), ("nv12", "yuvj420p"), ("yuvj420p", "nv21"), ("yuvj420p", "p010le"), ("p010le", "yuvj420p"), ("yuvj440p", "yuv444p12le"),
         ("yuv420p", "gray8"), ("nv12", "gray8"), ("yuv420p10le", "gray8"), ("yuv444p", "gray10le"), ("yuvj420p", "gray8"), ("yuv420p", "yuvj411p"),
         ("yuvj420p", "yuyv422"), ("yuv420p", "uyvy422")]
GEOMS = [(644, 70, 322, 35, SWS_BILINEAR), (400, 66, 330, 54, SWS_BICUBIC), (640, 48, 640, 48, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (1284, 36, 428, 12, SWS_BICUBIC)]


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_range_pairs(form, pair):
    sfmt, dfmt = pair
    for (sw, sh, dw, dh, fl) in GEOMS:
        opts = None
        if dfmt in ("yuyv422", "uyvy422") and "yuvj" not in sfmt:
            opts = dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
        path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=FORMS[form], opts=opts)
        if form != "old":
            assert "strip" in path or "plane1" in path or path.startswith("unscaled:"), (pair, sw, dw, path)   # (two full-range sides at one size: planarCopy)


@pytest.mark.parametrize("fl", [SWS_POINT, SWS_AREA, SWS_BILINEAR, SWS_BICUBIC, SWS_GAUSS, SWS_LANCZOS, SWS_BICUBIC | SWS_ACCURATE_RND])
def test_scalers_with_range_options(fl):
    """src_range / dst_range as context options on formats whose names say nothing, both directions, shifted chroma positions"""
    for (sr, dr) in ((0, 1), (1, 0)):
        for sfmt, dfmt in (("yuv420p", "yuv420p"), ("yuv444p", "yuv420p"), ("nv12", "nv12"), ("yuv420p10le", "yuv420p"), ("yuv422p", "yuv444p10le")):
            for (sw, sh, dw, dh) in ((1280, 72, 640, 36), (900, 40, 449, 33), (640, 36, 960, 54), (2600, 20, 1000, 10), (1920, 64, 240, 8)):
                opts = dict(dither=1, src_range=sr, dst_range=dr, src_h_chr_pos=0, src_v_chr_pos=128, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw + sr, opts=opts, tune=T0)
                if sw != 1920:      # (8:1 with 4:4:4 -> 4:2:0 is a 16:1 chroma step: beyond the strip plans for the short scalers' sparse windows)
                    assert "strip" in path or "plane1" in path, (sfmt, dfmt, sw, dw, path)


def test_rgb_sources_into_full_range_yuv():
    """RGB sources are limited range after the readers (utils.c:877-880): into yuvj* the luma and chroma lines are converted ToJpeg"""
    for sfmt in ("rgb24", "bgra", "gbrp", "rgb565le", "x2rgb10le", "gbrp10le"):
        for dfmt in ("yuvj420p", "yuvj422p", "yuvj444p"):
            for (sw, sh, dw, dh, fl) in ((644, 40, 516, 32, SWS_BICUBIC), (640, 48, 320, 24, SWS_BILINEAR), (640, 32, 640, 32, SWS_BICUBIC), (320, 24, 644, 48, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw + len(sfmt), tune=T0)
                if sfmt in ("rgb24", "bgra", "gbrp") or fl != SWS_BILINEAR:   # (the per-kind reader's 16-bit lines at bilinear 2:1: windows of 65 chunks, no strip plan with or without a range change)
                    assert "strip" in path or "rgbsrc_unity" in path, (sfmt, dfmt, sw, dw, path)


def test_same_size_rgb_into_full_range_yuv_wave_march():
    """capture -> JPEG / MJPEG encoder at the same size: sws_k_rgbsrc_unity2's RNG instantiations (every ring depth: bicubic / bilinear / Lanczos chroma
    filters, 4:2:0 / 4:2:2 / semi-planar destinations), widths that are not multiples of 4 and unaligned frames keep other routes"""
    for sfmt in ("rgb24", "bgr24", "bgra", "argb", "gbrp"):
        for dfmt in ("yuvj420p", "yuvj422p"):
            for fl in (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT):
                for (w, h) in ((640, 48), (1284, 34), (256, 7)):
                    path, _ = run_case(w, h, sfmt, w, h, dfmt, fl | BX, seed=w + len(sfmt), tune=T0)
                    assert path == "main:rgbsrc_unity", (sfmt, dfmt, w, path)
    opts = dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
    for dfmt in ("nv12", "nv21", "yuv420p"):
        path, _ = run_case(1280, 64, "bgra", 1280, 64, dfmt, SWS_BICUBIC | BX, seed=3, opts=opts, tune=T0)
        assert path == "main:rgbsrc_unity", (dfmt, path)
    run_case(642, 32, "rgb24", 642, 32, "yuvj420p", SWS_BICUBIC | BX, seed=4, tune=T0)
    run_case(640, 32, "rgb24", 640, 32, "yuvj420p", SWS_BICUBIC | BX, seed=5, tune=dict(T0, no_rgbsrc2=1))
    run_case(640, 32, "rgb24", 640, 32, "yuvj420p", SWS_BICUBIC | BX, seed=6, device_frames=False)


def test_packed_422_sources_and_alpha_planes():
    """yuyv422 / uyvy422 sources (MJPEG-less cameras) into full range; yuva420p -> yuva420p with a range change: the alpha plane is scaled but NOT converted"""
    for sfmt in ("yuyv422", "uyvy422"):
        for (sw, sh, dw, dh) in ((640, 48, 426, 32), (640, 48, 640, 48)):
            run_case(sw, sh, sfmt, dw, dh, "yuvj420p", SWS_BICUBIC | BX, seed=dw, tune=T0)
    for (sr, dr) in ((0, 1), (1, 0)):
        opts = dict(dither=1, src_range=sr, dst_range=dr, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
        for sfmt, dfmt in (("yuva420p", "yuva420p"), ("yuva444p", "yuva420p"), ("yuva420p10le", "yuva420p")):
            run_case(640, 48, sfmt, 320, 24, dfmt, SWS_BICUBIC | BX, seed=5 + sr, opts=opts, tune=T0)
            run_case(640, 48, sfmt, 800, 60, dfmt, SWS_BILINEAR | BX, seed=7 + sr, opts=opts, tune=T0)


def test_full_size_camera_and_thumbnail_shapes():
    """1080p MJPEG camera frame -> encoder input, 4K -> 720p thumbnail into JPEG range, from HBM and from host frames"""
    for devf in (True, False):
        run_case(1920, 1080, "yuvj422p", 1920, 1080, "yuv420p", SWS_BICUBIC | BX, seed=11, device_frames=devf)
        run_case(3840, 2160, "yuv420p", 1280, 720, "yuvj420p", SWS_BICUBIC | BX, seed=12, device_frames=devf)
    run_case(1920, 1080, "yuv420p", 320, 180, "yuvj420p", SWS_BICUBIC | BX, seed=13)
    run_case(1920, 1080, "bgra", 1920, 1080, "yuvj420p", SWS_BICUBIC | BX, seed=14)
    run_case(1920, 1080, "yuv420p", 640, 360, "gray8", SWS_BILINEAR | BX, seed=15)


def test_gray_sources_into_yuv():
    """gray -> planar / semi-planar YUV (round 5): the strip kernels' luma launch (gray8 is full range: a range conversion into limited-range YUV) and
    sws_k_gray_chroma for the chroma planes -- the reference's chroma writers over constant lines (ff_init_desc_no_chr), with the real vertical bank
    and the dither of sources beyond 8 bits"""
    for sfmt in ("gray8", "gray10le", "gray12le", "gray16le", "gray10be"):
        for dfmt in ("yuv420p", "yuv422p", "yuv444p", "nv12", "nv21", "yuv420p10le", "p010le", "yuv444p12le", "yuv420p16le", "p016le", "yuvj420p", "yuv410p"):
            for (sw, sh, dw, dh, fl) in ((644, 70, 324, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (640, 48, 640, 48, SWS_BICUBIC),
                                         (640, 3, 320, 24, SWS_BICUBIC), (1284, 36, 428, 12, SWS_LANCZOS)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0)
                if (sw, dw) == (400, 332) and dfmt not in ("yuv420p16le", "p016le"):     # (19-bit lines with a range conversion: 64-bit arithmetic, the old kernels)
                    assert "strip" in path, (sfmt, dfmt, path)
    opts = dict(dither=1, src_range=0, dst_range=1, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=0, dst_v_chr_pos=128, threads=1)
    run_case(640, 48, "gray8", 320, 24, "yuv420p", SWS_BICUBIC | BX, seed=3, opts=opts, tune=T0)
    run_case(1920, 1080, "gray8", 1280, 720, "nv12", SWS_BICUBIC | BX, seed=4)
    run_case(1920, 1080, "gray16le", 960, 540, "yuv420p10le", SWS_BILINEAR | BX, seed=5, device_frames=False)


def test_gray_sources_into_yuv_at_the_same_size():
    """a monochrome camera into an encoder (round 5): identity luma filters -- the mixed plan's streaming plane pass (with the range conversion for 8-bit planes) and
    sws_k_gray_chroma(_vec) alone, no strip plan; ragged widths (the thread behind the last whole 16-byte group), the scalar chroma kernel (no_wave), host frames,
    limited-range gray (no range conversion), dither of sources beyond 8 bits"""
    P = "main:plane1+gray_chroma"
    for sfmt, dfmt, want in (("gray8", "yuv420p", P), ("gray8", "nv12", P), ("gray8", "nv21", P), ("gray8", "yuv422p", P), ("gray8", "yuv444p", P), ("gray8", "yuvj420p", None), ("gray8", "yuv410p", P),
                             ("gray10le", "yuv420p10le", None), ("gray10le", "yuv420p", None), ("gray16le", "p010le", None), ("gray12le", "yuv444p12le", None), ("gray8", "yuv420p10le", None),
                             ("gray8", "p010le", None), ("gray10le", "nv12", None), ("gray8", "yuv420p16le", None)):
        for (w, h) in ((640, 48), (1920, 1080), (642, 37), (1366, 50), (30, 9), (18, 2)):
            for tune in (None, dict(no_wave=1)):
                path, _ = run_case(w, h, sfmt, w, h, dfmt, SWS_BICUBIC | BX, seed=w + len(dfmt), tune=tune)
                if want:
                    assert path == want, (sfmt, dfmt, w, h, path)
        run_case(1280, 720, sfmt, 1280, 720, dfmt, SWS_BILINEAR | BX, seed=2, device_frames=False)
        opts = dict(dither=1, src_range=0, dst_range=0, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
        run_case(644, 36, sfmt, 644, 36, dfmt, SWS_BICUBIC | BX, seed=3, opts=opts)
    assert run_case(640, 48, "gray8", 640, 48, "yuv420p", SWS_BICUBIC | BX, tune=dict(no_mixed=1))[0] != P


def test_rgb_sources_into_gray_and_ranges_with_19_bit_lines():
    """RGB -> gray (frames for analysis: the reader pre-pass's luma plane under the luma launch alone; gray is full range, RGB's lines limited: ToJpeg) and
    the 19-bit range conversion (lum / chrRangeToJpeg16_c ...: 64-bit arithmetic) in sws_k_strip_wide: YUV -> gray16, gray -> 16-bit YUV, yuvj -> 16-bit YUV"""
    for sfmt in ("rgb24", "bgra", "gbrp", "rgb565le", "x2rgb10le", "rgb48le"):
        for dfmt in ("gray8", "gray10le", "gray16le"):
            for (sw, sh, dw, dh, fl) in ((644, 40, 516, 32, SWS_BICUBIC), (640, 48, 320, 24, SWS_BILINEAR), (640, 32, 640, 32, SWS_BICUBIC), (320, 24, 644, 48, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=dw + len(sfmt), tune=T0)
                if (sw, dw) == (644, 516):
                    assert "rgbread" in path, (sfmt, dfmt, path)
    for sfmt, dfmt in (("yuv420p", "gray16le"), ("nv12", "gray16le"), ("yuv420p10le", "gray16le"), ("gray8", "yuv420p16le"), ("gray16le", "p016le"), ("yuvj420p", "yuv420p16le"),
                       ("yuv420p", "yuv444p16le"), ("yuvj444p", "gbrpf32le"), ("rgb24", "yuv420p16le")):
        for (sr, dr) in ((None, None), (0, 1), (1, 0)):
            opts = None if sr is None else dict(dither=1, src_range=sr, dst_range=dr, src_h_chr_pos=-513, src_v_chr_pos=-513, dst_h_chr_pos=-513, dst_v_chr_pos=-513, threads=1)
            for (sw, sh, dw, dh, fl) in ((644, 70, 324, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0, opts=opts)
                if sw == 400:
                    assert "strip" in path, (sfmt, dfmt, sr, path)


def test_gray_sources_into_packed_rgb():
    """gray -> 24 / 32 bpp RGB through the LUT writers (round 5): luma launch (raw sums), sws_k_gray_chroma writes the chroma SUMS of the reference's constant chroma lines,
    sws_k_fullchr_rgb finishes (a gray source counts as 4:4:4: forced full chroma)"""
    from librempeg_amd import SWS_FULL_CHR_H_INT
    for sfmt in ("gray8", "gray10le", "gray16le"):
        for dfmt in ("bgra", "rgb24", "argb", "bgr24", "rgb0"):
            for (sw, sh, dw, dh, fl) in ((644, 70, 324, 35, SWS_BILINEAR), (400, 66, 332, 54, SWS_BICUBIC), (320, 40, 640, 80, SWS_BICUBIC), (640, 48, 640, 48, SWS_BICUBIC),
                                         (640, 3, 320, 24, SWS_BICUBIC), (640, 48, 320, 48, SWS_BICUBIC), (400, 66, 331, 54, SWS_BICUBIC)):
                path, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, fl | BX, seed=sw + len(dfmt), tune=T0)
                if (sw, dw) == (400, 332):
                    assert path.endswith("+fullchr_rgb"), (sfmt, dfmt, path)     # (a gray source counts as 4:4:4: full chroma is forced, utils.c:1277-1285)
            run_case(400, 66, sfmt, 332, 54, dfmt, SWS_BICUBIC | SWS_FULL_CHR_H_INT | BX, seed=9, tune=T0)
            # (the epilogue computes the constant chroma sums itself; no_wave: sws_k_gray_chroma writes them into planes as before)
            for (sw, sh, dw, dh) in ((400, 66, 332, 54), (640, 48, 640, 48), (644, 70, 324, 35), (640, 3, 320, 24), (322, 31, 322, 31)):
                run_case(sw, sh, sfmt, dw, dh, dfmt, SWS_LANCZOS | BX, seed=sw + 1, tune=dict(T0, no_wave=1))
                run_case(sw, sh, sfmt, dw, dh, dfmt, SWS_LANCZOS | BX, seed=sw + 2, tune=T0)
    run_case(1920, 1080, "gray8", 1280, 720, "bgra", SWS_BICUBIC | BX, seed=4)
    run_case(3840, 2160, "gray16le", 3840, 2160, "bgra", SWS_BICUBIC | BX, seed=5)
    run_case(1920, 1080, "gray10le", 1920, 1080, "rgb24", SWS_BICUBIC | BX, seed=6, device_frames=False)
